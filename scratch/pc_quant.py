"""PointConv level 0 (d = 8, train mode): forward and forward + backward time against the number of points -- do the kernels pay
for wave-slot quantization (uvstats<8> 4 wavefronts / SIMD, bwd_input<8> 3, bwd_params<8> 2; 4 x 40960 points = 5 wavefronts / SIMD)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda', 0)
for N in (8192, 16384, 24576, 32768, 40960, 49152, 65536, 81920):
    gen = torch.Generator().manual_seed(1234)
    data, _ = bench.make_batch(0, 4, N, dev, gen, 'morton')
    r = bench.roofline_pointconv(data, dev, 8)
    print('N %6d  waves/SIMD %.2f  fwd %6.2f us  fwd+bwd %7.2f us   per 1k points fwd %.4f  fwd+bwd %.4f' %
          (N, 4 * N / 32 / 1024, r['avg_launch_us'], r['fwd_bwd_us'], r['avg_launch_us'] / (4 * N / 1000), r['fwd_bwd_us'] / (4 * N / 1000)), flush=True)
