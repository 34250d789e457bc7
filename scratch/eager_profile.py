"""cProfile of the reference's five-line step run EAGERLY on the HIP modules (trainval.py:99-106): where the ~27 us of host time per
launch go.  usage: python3 scratch/eager_profile.py > gpurun_out/eager_profile.txt"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import torch.nn.functional as F
from crfconv_amd import models
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
torch.manual_seed(0)
net = models.PointConvBig(6, 13, use_crf=True, steps=3).to(dev).train()
cw = torch.ones(13, device=dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)


def one():
    opt.zero_grad()
    loss = F.cross_entropy(net(data), data.y.reshape(-1) - 1, weight=cw, ignore_index=-1)
    loss.backward()
    opt.step()
    return loss.detach()


for _ in range(5):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    one()
torch.cuda.synchronize()
print('eager ms per step: %.3f' % ((time.perf_counter() - t0) / 20 * 1e3))
if os.environ.get('PROFILE_BACKWARD'):
    # the engine's device thread runs the backward functions: on the calling thread instead, so that the profile sees them
    torch.autograd.set_multithreading_enabled(False)
    one()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    one()
torch.cuda.synchronize()
pr.disable()
if os.environ.get('PROFILE_CALLERS'):
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats('tottime').print_callers('torch.empty|empty_like|zeros')
    print(s.getvalue()[:12000])
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
