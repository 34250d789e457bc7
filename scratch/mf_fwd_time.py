"""Level-0 mean-field forward, HIP-event time (bench.roofline_meanfield)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
for rep in range(3):
    r = bench.roofline_meanfield(data, dev, 8, 3)
    print('forward avg %.2f us  min %.2f us  frac %.4f' % (r['avg_launch_us'], r['min_launch_us'], r['frac']), flush=True)
