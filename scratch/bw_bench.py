"""HBM rates of this device for pure writes, pure reads and copies (framework kernels, graph-replayed): the ceilings the
streaming kernels are priced against."""
import torch
dev = torch.device('cuda', 0)
for mb in (21, 84, 336):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    for name, fn, bytes_ in (('fill (write only)', lambda: a.fill_(1.0), 4 * n), ('sum (read only)', lambda: a.sum(), 4 * n),
                             ('copy (read + write)', lambda: b.copy_(a), 8 * n), ('scale in place (read + write)', lambda: a.mul_(1.0001), 8 * n)):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): g.replay()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 100 * 1e3
        print('%4d MB  %-30s %7.1f us  %5.2f TB/s' % (mb, name, us, bytes_ / us / 1e6), flush=True)
