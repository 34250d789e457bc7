for v in "" lf192 lf384 lf512; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export CRFCONV_LIB=$GRAFT_REPO_ROOT/scratch/variants/lib_$v.so; else unset CRFCONV_LIB; fi
  timeout -k 10 200 python3 scratch/lin_bench.py 2>&1 | grep -v amdgpu | grep "stats 1" | head -8
  timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=[json.loads(l) for l in sys.stdin if l.startswith('{')][0]
print('step %.3f ms  value %.2f' % (r['ms_per_step'], r['value']))"
done
