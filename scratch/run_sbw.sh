for v in 128 100000 64; do
  echo "== CRFCONV_SMALL_BWD_WIDE_FROM=$v"
  CRFCONV_SMALL_BWD_WIDE_FROM=$v timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=[json.loads(l) for l in sys.stdin if l.startswith('{')][0]
print('step %.3f ms  value %.2f' % (r['ms_per_step'], r['value']))"
done
