"""Does the level-0 first kernel pay for wave-slot quantization?  sim_step_fast<8,16> holds 104 VGPRs = 4 wavefronts per SIMD; 4 x 40960
points are 5120 wavefronts = 5 per SIMD.  Forward time against the number of points (whole wavefront rounds at 4 / 8 / 12 per SIMD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda', 0)
for N in (24576, 32768, 36864, 40960, 45056, 49152, 65536):
    gen = torch.Generator().manual_seed(1234)
    data, _ = bench.make_batch(0, 4, N, dev, gen, 'morton')
    for T in (1, 3):
        f = bench.roofline_meanfield(data, dev, 8, T, level=0)
        print('N %6d  m %7d  waves/SIMD %.2f  T %d  fwd %6.2f us (min %6.2f)  per 1k points %.4f us' %
              (N, 4 * N, 4 * N / 32 / 1024, T, f['avg_launch_us'], f['min_launch_us'], f['avg_launch_us'] / (4 * N / 1000)), flush=True)
