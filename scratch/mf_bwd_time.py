"""Level-0 mean-field backward, HIP-event time (bench.roofline_meanfield_bwd) + parity against the step-by-step launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
for rep in range(2):
    r = bench.roofline_meanfield_bwd(data, dev, 8, 3)
    print('backward avg %.2f us  min %.2f us' % (r['avg_launch_us'], r['min_launch_us']), flush=True)
