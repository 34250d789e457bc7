for rep in 1 2; do
for v in 12288 65536; do
  CRFCONV_MFMA_MIN_ROWS=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
r=[json.loads(l) for l in sys.stdin if l.startswith('{')][0]
print('MIN_ROWS=$v step %.3f ms  value %.2f  pipelined %.3f' % (r['ms_per_step'], r['value'], r['pipelined_ms_per_batch']))"
done
done
