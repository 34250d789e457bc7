for rep in 1 2; do
for v in "" "CRFCONV_NO_APPLY_FROM_RECORDS=1"; do
  env $v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
r=[json.loads(l) for l in sys.stdin if l.startswith('{')][0]
print('[$v] step %.3f ms  value %.2f' % (r['ms_per_step'], r['value']))"
done
done
