"""-m gpu: the batch-sharded training step on more than one rank.  No multi-GPU box is available to these tests, so
two ranks SHARE the one GPU and talk over gloo (CRFCONV_DIST_BACKEND=gloo, the documented test path of
distributed.init_from_env); the code under test -- FlatGradAllReduce.pack / allreduce_mean, FlatSGD -- is the code
bench.py runs over RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import _seeded as S
from gpu_util import DEV, assert_close, t

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests', 'golden'))
import numpy as np, torch
import _seeded as S
import crfconv_amd
from crfconv_amd import distributed as D, models, ops, optim
rank, world, local = D.init_from_env()
dev = torch.device('cuda', local)
torch.cuda.set_device(dev)
B, N = 2, 2048
pos = np.stack([S.make_cloud(300 + b, N, box=(2, 2, 1)) for b in range(B)])
feats = np.concatenate([pos, S.uniform(300, 'rgb', (B, N, 3), 0, 1)], -1).astype(np.float32)
labels = S.integers(300, 'y', (B, N), 1, 14)
choices, n = [], N
for i, r in enumerate((4, 4, 4, 2, 2)):
    choices.append(torch.from_numpy(S.permutation(300, 'c%%d' %% i, n)[: n // r]))
    n //= r
def batch(sl):
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).to(dev)
    return crfconv_amd.multiscale_compute(f(pos), x=f(feats), y=f(labels), choices=choices, ratio=(4, 4, 4, 2, 2))
torch.manual_seed(11)                          # the constructor's own initialisation (what training starts from), same on
net = models.PointConvBig(6, 13, True, 3)      # both ranks; a random-GAIN state dict makes the CRF soft-max so sensitive
net = net.to(dev).eval()                       # that vendor-GEMM rounding at the 16-row level moves gradients by percents
# eval mode = running-statistics BatchNorm: shards and the big batch see the same network
def grads_of(data):
    for p in net.parameters():
        p.grad = None
    loss = ops.training_loss(net(data), data.y, None, ignore_index=-1)
    loss.backward()
    return loss
# reference: the un-sharded batch on this rank
grads_of(batch(slice(0, B)))
ref = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.requires_grad]).clone()
# sharded: one cloud per rank, ONE flat all-reduce, then the SGD step on the flat vectors
bucket = D.FlatGradAllReduce(net)
opt = optim.FlatSGD(bucket, lr=0.01, momentum=0.9, weight_decay=1e-4)
before = opt.flat.clone()
opt.zero_grad()
grads_of(batch(slice(rank, rank + 1)))
bucket.allreduce_mean()
opt.step()
torch.cuda.synchronize()
torch.save({'flat': bucket.flat.cpu(), 'ref': ref.cpu(), 'before': before.cpu(), 'after': opt.flat.cpu()}, os.environ['OUT'] + '.%%d' %% rank)
D.dist.barrier()
D.dist.destroy_process_group()
'''


def test_two_ranks_sharing_the_gpu_match_the_big_batch(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, OUT=str(tmp_path / 'g'), MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', WORLD_SIZE='2',
               CRFCONV_DIST_BACKEND='gloo', OMP_NUM_THREADS='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0, r1 = (torch.load(str(tmp_path / ('g.%d' % r))) for r in range(2))
    assert torch.equal(r0['flat'], r1['flat'])                                 # replicas hold identical averaged gradients
    scale = max(1.0, float(r0['ref'].abs().max()))
    assert float((r0['flat'] - r0['ref']).abs().max()) <= 1e-4 * scale          # == gradient of the un-sharded batch
    assert torch.equal(r0['after'], r1['after'])                               # and take the same SGD step
    want = r0['before'] - 0.01 * (r0['flat'] + 1e-4 * r0['before'])            # first step: buf = g + wd p
    assert float((r0['after'] - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))


def test_single_rank_documented_step_moves_the_parameters():
    """INTEGRATION.md's loop  zero_grad -> backward -> allreduce_mean -> step  on ONE rank (no process group): the
    gradients must reach the flat bucket and the parameters must move by -lr * grad (round-1 ADVICE: allreduce_mean used
    to return early at world size 1 and the step then saw zeros)."""
    from crfconv_amd import optim
    from crfconv_amd.distributed import FlatGradAllReduce
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(5, 9), torch.nn.Linear(9, 3)).to(DEV)
    bucket = FlatGradAllReduce(net)
    opt = optim.FlatSGD(bucket, lr=0.1)
    x = torch.randn(16, 5, device=DEV)
    before = opt.flat.clone()
    opt.zero_grad()
    net(x).square().mean().backward()
    g = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    assert float(g.abs().max()) > 0
    bucket.allreduce_mean()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
    opt.step()
    assert_close(opt.flat, before - 0.1 * g, 1e-6, 'p - lr g')


def test_flat_sgd_follows_exponential_lr_scheduler_eager_and_captured():
    """trainval.py:69-73: SGD wrapped by ExponentialLR.  FlatSGD is a torch.optim.Optimizer (param_groups / lr), the
    scheduler drives it, and a CAPTURED step honours the new rate after push_hyper() because the kernel reads the
    hyper-parameters from device memory."""
    from crfconv_amd import optim
    from crfconv_amd.distributed import FlatGradAllReduce
    torch.manual_seed(2)
    net = torch.nn.Linear(6, 4).to(DEV)
    ref = torch.nn.Linear(6, 4).to(DEV)
    ref.load_state_dict(net.state_dict())
    bucket = FlatGradAllReduce(net)
    opt = optim.FlatSGD(bucket, lr=0.05, momentum=0.9, weight_decay=1e-3)
    assert isinstance(opt, torch.optim.Optimizer)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3)
    sch, rsch = (torch.optim.lr_scheduler.ExponentialLR(o, gamma=0.5) for o in (opt, ropt))
    graph = None
    for step in range(5):
        grads = [t(S.uniform(step, 'h%d' % i, tuple(p.shape))) for i, p in enumerate(ref.parameters())]
        for p, v, g in zip(ref.parameters(), bucket.views, grads):
            p.grad = g.clone()
            v.copy_(g)
        if step < 2:
            opt.step()
        else:
            if graph is None:                                 # capture ONE step; the following ones are replays
                snap, msnap = opt.flat.clone(), opt.buf.clone()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    opt.step()
                torch.cuda.current_stream().wait_stream(side)
                with torch.no_grad():
                    opt.flat.copy_(snap); opt.buf.copy_(msnap)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    opt.step()
                with torch.no_grad():
                    opt.flat.copy_(snap); opt.buf.copy_(msnap)
            graph.replay()
        ropt.step()
        sch.step(); rsch.step()
        opt.push_hyper()
        assert abs(opt.lr - ropt.param_groups[0]['lr']) < 1e-12
        for a, b in zip(net.parameters(), ref.parameters()):
            assert_close(a, b, 1e-6, 'step %d' % step)


def test_bench_pipelined_loop_on_two_ranks(tmp_path):
    """bench.py as the driver launches it for N = 2 (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment), here with both ranks on the one GPU over gloo and a small batch: the captured step with the flat all-reduce
    between its two graphs AND the pipelined loop (CollatePipeline: every iteration collates a fresh batch on a side stream
    and trains on it) must run to the end on both ranks and rank 0 must print the one JSON line."""
    import json
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', WORLD_SIZE='2', CRFCONV_DIST_BACKEND='gloo',
               OMP_NUM_THREADS='2')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--batch', '2',
           '--points', '8192', '--no-cpu-baseline']
    outs = [open(str(tmp_path / ('out.%d' % r)), 'w') for r in range(2)]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=outs[r], stderr=subprocess.STDOUT, cwd=ROOT)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0, open(str(tmp_path / 'out.0')).read()[-2000:]
    lines = [l for l in open(str(tmp_path / 'out.0')).read().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['value'] > 0 and rec['scaling'] == 'weak'
    assert rec['pipelined_ms_per_batch'] is not None and rec['pipelined_ms_per_batch'] > 0
    assert rec['config']['global_batch'] == 4
    assert not [l for l in open(str(tmp_path / 'out.1')).read().splitlines() if l.startswith('{')]      # rank 0 alone reports


def test_bench_self_launch_one_rank_over_rccl(tmp_path):
    """`python bench.py --gpus 1 --spawn` WITHOUT a launcher environment: bench.py itself starts its rank as a child process
    (before it imports torch or touches the GPU), the child builds the one-rank process group over the `nccl` backend (= RCCL)
    and runs the whole benchmark through the grouped code path -- the flat all-reduce between the two captured graphs, its own
    HIP-event time in the line -- and the parent relays the JSON line and the exit status.  The shape of the command the driver
    uses for `--gpus N` on a multi-GPU node, at the one N this box can run."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'CRFCONV_DIST_BACKEND')}
    env['OMP_NUM_THREADS'] = '4'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--spawn', '--steps', '4', '--warmup', '1', '--batch', '2',
           '--points', '8192', '--no-cpu-baseline']
    with open(str(tmp_path / 'out'), 'w') as fo, open(str(tmp_path / 'err'), 'w') as fe:
        rc = subprocess.run(cmd, env=env, stdout=fo, stderr=fe, cwd=ROOT, timeout=900).returncode
    assert rc == 0, open(str(tmp_path / 'err')).read()[-3000:]
    lines = [l for l in open(str(tmp_path / 'out')).read().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 1 and rec['rccl_ranks_seen'] == 1 and rec['dist_backend'] == 'nccl'
    assert rec['allreduce_us'] is not None and rec['allreduce_us'] > 0 and rec['value'] > 0
    assert rec['trainval_eager_ms_per_step'] > 0 and rec['trainval_captured_ms_per_step'] > 0
    # a failing rank must surface as a non-zero exit status of the parent
    bad = subprocess.run(cmd + ['--points', '-5'], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT, timeout=600)
    assert bad.returncode != 0


def test_bench_self_launch_two_ranks_sharing_the_gpu(tmp_path):
    """`python bench.py --gpus 2` as the driver would type it on a multi-GPU node -- no launcher, bench.py starts both ranks itself -- here
    with the two ranks on the one GPU over gloo (CRFCONV_DIST_BACKEND, inherited by the children): both ranks must finish, the parent must
    relay exactly one JSON line with n_gpus = 2 and the group's size and backend in it."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(CRFCONV_DIST_BACKEND='gloo', OMP_NUM_THREADS='2')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '2', '--points', '8192',
           '--no-cpu-baseline']
    with open(str(tmp_path / 'out'), 'w') as fo, open(str(tmp_path / 'err'), 'w') as fe:
        rc = subprocess.run(cmd, env=env, stdout=fo, stderr=fe, cwd=ROOT, timeout=900).returncode
    assert rc == 0, open(str(tmp_path / 'err')).read()[-3000:]
    lines = [l for l in open(str(tmp_path / 'out')).read().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['rccl_ranks_seen'] == 2 and rec['dist_backend'] == 'gloo' and rec['scaling'] == 'weak'
    assert rec['config']['global_batch'] == 4 and rec['value'] > 0 and rec['allreduce_us'] > 0


GUARD_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch
import crfconv_amd
from crfconv_amd import _lib, distributed as D, ops, optim
rank, world, local = D.init_from_env()
dev = torch.device('cuda', local)
torch.cuda.set_device(dev)
torch.manual_seed(3)
net = torch.nn.Linear(9, 17).to(dev)
D.broadcast_parameters(net)
bucket = D.FlatGradAllReduce(net)
opt = optim.FlatSGD(bucket, lr=0.1, momentum=0.9, weight_decay=1e-4, check_every=0, grad_scale=1.0 / world)
ws = ops.gridsync_ws(dev)
word = _lib.load().crfconv_gridsync_fail_word()
def one_step(value):
    opt.zero_grad()
    for p in net.parameters():
        p.grad = torch.full_like(p, value)
    bucket.allreduce_sum()
    opt.step()
one_step(1.0)                                   # a healthy step: momentum is non-zero afterwards
p0, b0 = opt.flat.clone(), opt.buf.clone()
if rank == 1:
    ws[word] = 0x101                            # rank 1's one-launch kernel "timed out": its gradient is NaN
for _ in range(2):
    one_step(float('nan') if rank == 1 else 1.0)
torch.cuda.synchronize()
kept = bool(torch.equal(opt.flat, p0) and torch.equal(opt.buf, b0))
raised = False
try:
    ops.check_gridsync(dev, reduced_flag=bucket.guard)
except _lib.CrfConvError:
    raised = True
one_step(1.0)                                   # flags cleared on both ranks: the update runs again, from intact state
torch.cuda.synchronize()
moved = bool(torch.isfinite(opt.flat).all()) and not torch.equal(opt.flat, p0)
torch.save({'kept': kept, 'raised': raised, 'moved': moved, 'flat': opt.flat.cpu()}, os.environ['OUT'] + '.%%d' %% rank)
D.dist.barrier()
D.dist.destroy_process_group()
'''


def test_a_barrier_failure_on_one_rank_stops_the_update_on_every_rank(tmp_path):
    """the failed rank's NaN gradient is summed into every rank's bucket by the all-reduce, so the guard must be
    global: the flag travels as one more slot of the bucket (FlatGradAllReduce.publish_guard), every rank's update kernel skips
    on the reduced slot, and check_gridsync(reduced_flag=...) raises on every rank in the same step."""
    script = tmp_path / 'worker.py'
    script.write_text(GUARD_WORKER % ROOT)
    env = dict(os.environ, OUT=str(tmp_path / 'g'), MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', WORLD_SIZE='2',
               CRFCONV_DIST_BACKEND='gloo', OMP_NUM_THREADS='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0, r1 = (torch.load(str(tmp_path / ('g.%d' % r))) for r in range(2))
    for r in (r0, r1):
        assert r['kept'] and r['raised'] and r['moved'], r
    assert torch.equal(r0['flat'], r1['flat'])                  # replicas never diverged


def test_an_eager_update_sees_the_failure_word_of_the_captured_graph_buffer():
    """captured kernels poison the 'capture' barrier workspace; an EAGER optimizer.step() behind them reads the
    current stream's -- the guard now looks at every workspace of the device."""
    from crfconv_amd import _lib, ops, optim
    from crfconv_amd.distributed import FlatGradAllReduce
    dev = torch.device('cuda', 0)
    net = torch.nn.Linear(5, 7).to(DEV)
    bucket = FlatGradAllReduce(net)
    opt = optim.FlatSGD(bucket, lr=0.1, momentum=0.9, check_every=0)
    ops.gridsync_ws(dev)
    ops.check_gridsync(dev)
    cap = ops._sync_ws[(0, 'capture')]
    word = _lib.load().crfconv_gridsync_fail_word()
    was = ops.state.small_mlp_disabled
    try:
        bucket.flat.fill_(1.0)
        opt.step()
        p0 = opt.flat.clone()
        cap[word] = 0x102
        bucket.flat.fill_(float('nan'))
        opt.step()
        torch.cuda.synchronize()
        assert torch.equal(opt.flat, p0)
        with pytest.raises(_lib.CrfConvError, match='grid barrier timed out'):
            ops.check_gridsync(dev)
        bucket.guard.fill_(2.0)                                  # the reduced slot alone stops the update too
        bucket.flat.fill_(1.0)
        opt.step()
        torch.cuda.synchronize()
        assert torch.equal(opt.flat, p0)
        with pytest.raises(_lib.CrfConvError, match='another rank'):
            ops.check_gridsync(dev, reduced_flag=bucket.guard)
        opt.step()
        torch.cuda.synchronize()
        assert not torch.equal(opt.flat, p0)
    finally:
        ops.state.small_mlp_disabled = was


VOTE_WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests', 'golden'))
import numpy as np, torch
import _seeded as S
import crfconv_amd
from crfconv_amd import distributed as D, models
from crfconv_amd.sampling import PossibilitySampler, VoteAccumulator, vote_scene
rank, world, local = D.init_from_env()
dev = torch.device('cuda', local)
torch.cuda.set_device(dev)
n, crop, n_crops, C = 24000, 6000, 6, 8
rng = np.random.default_rng(11)
pts = (rng.random((n, 3)) * np.array([4.0, 4.0, 2.0])).astype(np.float32)
rgb = rng.random((n, 3)).astype(np.float32)
poss0 = np.random.default_rng(5).standard_normal(n) * 1e-3
net = models.PointConvBig(6, C, use_crf=True, steps=3)
net.load_state_dict(S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 9))
net = net.to(dev).eval()
def run(r, w):
    smp = PossibilitySampler([torch.from_numpy(pts).to(dev)], rgb=[torch.from_numpy(rgb).to(dev)], num_points=crop, split='test',
                             generator=torch.Generator().manual_seed(77), possibility=[poss0])
    votes = VoteAccumulator([n], C, device=dev, track_visits=True)
    vote_scene(smp, net, votes, n_crops, rank=r, world=w, generator=torch.Generator().manual_seed(3))
    return smp, votes
smp, votes = run(rank, world)                  # this rank's crops only (every rank draws the whole crop sequence)
votes.merge()                                  # all-gather + fold in rank order
# what ONE accumulator holds that ran rank 0's crops first, then rank 1's: folded locally from the two single-rank runs
parts = [run(r, world) for r in range(world)]
one = parts[0][1]
for r in range(1, world):
    one.fold_(parts[r][1].test_probs, parts[r][1].visits)
torch.cuda.synchronize()
torch.save({'merged': votes.test_probs[0].cpu(), 'visits': votes.visits[0].cpu(), 'one': one.test_probs[0].cpu(), 'one_visits': one.visits[0].cpu(),
            'poss': smp.possibility[0].cpu(), 'poss_full': parts[0][0].possibility[0].cpu()}, os.environ['OUT'] + '.%%d' %% rank)
D.dist.barrier()
D.dist.destroy_process_group()
'''


def test_scene_crops_sharded_over_two_ranks_merge_to_one_vote_table(tmp_path):
    """Config 5's sharding (crops of one scene over the ranks; SURVEY 8(e): inference needs no collective UNTIL the votes are read):
    `vote_scene(rank, world)` + `VoteAccumulator.merge()` on two gloo ranks sharing the GPU.  Both ranks end with the same table, equal
    to one accumulator that applied rank 0's crops first, then rank 1's; the sampler's possibilities evolve as on one GPU on every rank."""
    script = tmp_path / 'vote_worker.py'
    script.write_text(VOTE_WORKER % (ROOT, ROOT))
    env = dict(os.environ, OUT=str(tmp_path / 'v'), MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', WORLD_SIZE='2',
               CRFCONV_DIST_BACKEND='gloo', OMP_NUM_THREADS='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0, r1 = (torch.load(str(tmp_path / ('v.%d' % r))) for r in range(2))
    assert torch.equal(r0['merged'], r1['merged']) and torch.equal(r0['visits'], r1['visits'])
    assert torch.equal(r0['visits'], r0['one_visits']) and int(r0['visits'].sum()) == 6 * 6000
    assert torch.equal(r0['merged'], r0['one'])              # the same fold kernel on the same operands: bit for bit
    assert torch.equal(r0['poss'], r1['poss']) and torch.equal(r0['poss'], r0['poss_full'])
    assert float(r0['merged'].abs().max()) > 0
