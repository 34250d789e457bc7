"""Do independent branches of a captured hipGraph run concurrently on ROCm, and what does a fork/join cost?"""
import torch, time
dev = torch.device('cuda', 0)
def bench(g, n=200):
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for numel in (1 << 12, 1 << 20, 1 << 24):
    x = torch.randn(numel, device=dev); y = torch.randn(numel, device=dev)
    side = torch.cuda.Stream()
    for _ in range(3): x.mul_(1.0001); y.mul_(1.0001)
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        for _ in range(20): x.mul_(1.0001)
        for _ in range(20): y.mul_(1.0001)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for _ in range(20): y.mul_(1.0001)
        for _ in range(20): x.mul_(1.0001)
        main.wait_stream(side)
    g3 = torch.cuda.CUDAGraph()      # ten short forks: 2 main ops, 2 side ops, join
    with torch.cuda.graph(g3):
        main = torch.cuda.current_stream()
        for _ in range(10):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                y.mul_(1.0001); y.mul_(1.0001)
            x.mul_(1.0001); x.mul_(1.0001)
            main.wait_stream(side)
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4):
        for _ in range(20): x.mul_(1.0001)
    print('numel %9d: 40 ops in one chain %7.1f us | 20 + 20 on two branches %7.1f us | 10 x (2 || 2) fork-joins %7.1f us | 20 ops alone %7.1f us'
          % (numel, bench(g1), bench(g2), bench(g3), bench(g4)), flush=True)
