python3 -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for cfg in "default" "CRFCONV_BENCH_COLLATE_PRIORITY=0" "CRFCONV_BENCH_STEP_PRIORITY=-1" "CRFCONV_BENCH_STEP_PRIORITY=-1 CRFCONV_BENCH_COLLATE_PRIORITY=0"; do
  echo "== $cfg"
  if [ "$cfg" = "default" ]; then e=""; else e="$cfg"; fi
  env $e timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=[json.loads(l) for l in sys.stdin if l.startswith('{')][0]
print('step %.3f ms  pipelined %.3f ms  collate graph %.3f ms' % (r['ms_per_step'], r['pipelined_ms_per_batch'], r['preprocess_plus_refresh_graph_ms_per_batch']))"
done
