"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the nn.Module mirrors have the reference's state_dict layout, data containers behave,
the product refuses to run without a GPU (no CPU fallback), and the batch-sharded gradient
all-reduce is equivalent to one big batch (gloo, world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from crfconv_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'crfconv_amd.h')).read()
    declared = set(re.findall(r'\b(crfconv_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.crfconv_abi_version() == 1
    # argument validation happens before any HIP call, so it is testable without a GPU
    rc = lib.crfconv_meanfield_forward(None, None, None, 16, 1, 10, 7, None, None, 1, None, None, None)
    assert rc == -3
    rc = lib.crfconv_meanfield_forward_u16(None, None, None, None, 1, 1, 16, 1, 10, 7, None, None, 1, None, None, None)
    assert rc == -3 and b'H=7' in lib.crfconv_last_error()
    rc = lib.crfconv_knn_batch_dev(None, 1, 10, 3, None, 10, 4, None, None, None, 0, None)
    assert rc == -1


def test_state_dict_layout_matches_reference(golden):
    from crfconv_amd import models
    g = golden('g5_pointconvbig.npz')
    for tagc, use_crf in (('crf', True), ('ups', False)):
        net = models.PointConvBig(6, 13, use_crf=use_crf, steps=3)
        sd = net.state_dict()
        want = {k: (tuple(int(s) for s in sh.split(',')) if sh else ()) for k, sh in zip(g[tagc + '/keys'], g[tagc + '/shapes'])}
        assert set(sd) == set(want)
        for k, shp in want.items():
            assert tuple(sd[k].shape) == shp, k
    assert sum(p.numel() for p in models.PointConvBig(6, 13, True, 3).parameters()) == 820141
    crf = models.ContinuousGaussianCRFConv(64, 32, 32, steps=3)
    assert torch.equal(crf.c.data, torch.eye(8))                     # identity init (reference :36)
    g1 = golden('g1_crfconv.npz')
    crf.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g1.items() if k.startswith('sd/')}, strict=True)


def test_no_cpu_fallback():
    from crfconv_amd import models
    from crfconv_amd._lib import CrfConvError
    m = models.ContinuousGaussianCRFConv(64, 32, 32, steps=1)
    with pytest.raises(CrfConvError, match='no CPU path'):
        m(torch.zeros(2, 8, 64), torch.zeros(2, 32, 32), torch.zeros(2, 32, 1, dtype=torch.long),
          torch.zeros(2, 32, 16, dtype=torch.long))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'crfconv_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dirpath, f)
                assert '/root/reference' not in src, os.path.join(dirpath, f)


def test_data_containers():
    from crfconv_amd import Data, MultiScaleData
    d = MultiScaleData(x=torch.zeros(2, 4, 6), y=torch.ones(2, 4, dtype=torch.long),
                       multiscale=[Data(pos=torch.zeros(2, 4, 3), neighbor_idx=torch.zeros(2, 4, 2, dtype=torch.long))])
    e = d.to('cpu')
    assert e is not d and e.multiscale[0].pos.shape == (2, 4, 3) and 'x' in e.keys
    assert 'MultiScaleData' in repr(e)


def test_mlp_refuses_cpu_tensors():
    """MLP is on the path (every Linear runs on the MFMA kernels): like the CRF / PointConv layers it has no CPU
    route.  Its semantics against the oracle are checked on the GPU (tests/test_gpu_model.py)."""
    from crfconv_amd.models import MLP
    from crfconv_amd._lib import CrfConvError
    m = MLP(12, 20, activation=torch.nn.LeakyReLU(0.1))
    with pytest.raises(CrfConvError, match='no CPU path'):
        m(torch.zeros(3, 50, 12))


WORKER = r'''
import os, sys, torch
sys.path.insert(0, %r)
from crfconv_amd import distributed as D
rank, world, _ = D.init_from_env('gloo')
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.LeakyReLU(0.1), torch.nn.Linear(16, 5))
D.broadcast_parameters(net)
bucket = D.FlatGradAllReduce(net)
g = torch.Generator().manual_seed(7)
x = torch.randn(8, 6, generator=g); y = torch.randint(0, 5, (8,), generator=g)
shard = slice(rank * 4, rank * 4 + 4)
bucket.zero()
torch.nn.functional.cross_entropy(net(x[shard]), y[shard]).backward()
bucket.allreduce_mean()
torch.save(torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone(), os.environ['OUT'] + '.%%d' %% rank)
# the failure flag travels with the gradients: rank 1 alone publishes 1, both ranks must see a non-zero slot afterwards
assert float(bucket.guard) == 0.0 and bucket.wire.numel() == bucket.flat.numel() + 1
bucket.publish_guard = lambda: bucket.guard.fill_(float(rank))
bucket.allreduce_sum()
assert float(bucket.guard) == 1.0, float(bucket.guard)
if rank == 0:
    ref = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.LeakyReLU(0.1), torch.nn.Linear(16, 5))
    ref.load_state_dict(net.state_dict())
    torch.nn.functional.cross_entropy(ref(x), y).backward()
    torch.save(torch.cat([p.grad.reshape(-1) for p in ref.parameters()]), os.environ['OUT'] + '.ref')
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


def test_sharded_grad_allreduce_equals_big_batch(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, OUT=str(tmp_path / 'g'), MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', WORLD_SIZE='2',
               OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    g0, g1, ref = (torch.load(str(tmp_path / ('g.' + s))) for s in ('0', '1', 'ref'))
    assert torch.equal(g0, g1)                                   # replicas hold identical averaged grads
    assert float((g0 - ref).abs().max()) < 1e-6                  # == gradient of the un-sharded batch


def test_bn_counters_advance_as_one_vector():
    """ops.advance_counters: num_batches_tracked of all BatchNorms advance by one per training forward, survive
    state_dict round trips and module._apply (which breaks the shared storage and must be re-detected)."""
    import torch
    from crfconv_amd import ops
    net = torch.nn.Sequential(torch.nn.BatchNorm1d(4), torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    net[2].num_batches_tracked += 5
    for expect in (1, 2):
        with ops.advance_counters(net):
            ops.tick(net[0])                                   # inside the context the per-layer tick is a no-op
        assert int(net[0].num_batches_tracked) == expect and int(net[2].num_batches_tracked) == 5 + expect
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    assert sd['0.num_batches_tracked'].shape == () and int(sd['2.num_batches_tracked']) == 7
    net = net.double()                                         # _apply re-creates the buffers one by one
    with ops.advance_counters(net):
        pass
    assert int(net[0].num_batches_tracked) == 3 and int(net[2].num_batches_tracked) == 8
    net.load_state_dict(sd)
    with ops.advance_counters(net):
        pass
    assert int(net[0].num_batches_tracked) == 3 and int(net[2].num_batches_tracked) == 8
    ops.tick(net[0])
    assert int(net[0].num_batches_tracked) == 4


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                           # (__name__ != '__main__': no self-launch, no benchmark)
    return mod


def test_step_byte_model_reproduces_the_survey_figures():
    """bench.step_byte_model (the algorithmic bytes `roofline_step` is quoted on): the mean-field group must be SURVEY 8(d)'s
    81.9 MB forward for config 2, and its level-0 share the 46.5 MB of `roofline`."""
    bench = _load_bench()
    total, groups = bench.step_byte_model(4, 40960, 16, 3, 13)
    assert abs(groups['mean_field']['fwd'] - 81.9e6) < 0.1e6
    m, H, K, T = 4 * 40960, 8, 16, 3
    assert m * (4 * (K - 1) + 4 * H * (2 * T + 1)) == 46530560
    assert groups['mean_field']['bwd'] == 2 * groups['mean_field']['fwd'] + sum(4 * 40960 // 4 ** l * (4 * K + 4) for l in range(4))
    assert 3.3e9 < total < 3.7e9 and set(groups) == {'encoder_linear', 'pointconv', 'pool_gather', 'decoder_linear', 'mean_field', 'classifier_loss'}
    # scaling sanity: twice the points, twice the bytes
    assert abs(bench.step_byte_model(4, 81920, 16, 3, 13)[0] / total - 2.0) < 1e-9


def test_bench_starts_its_own_ranks_without_a_launcher_environment(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: N child processes with the torchrun variables on 127.0.0.1,
    the parent's exit status = the first failing rank's; under a launcher (WORLD_SIZE set) or at N = 1 without --spawn: nothing."""
    bench = _load_bench()
    started = []

    class FakeProc:
        def __init__(self, cmd, env):
            self.cmd, self.env, self.returncode = cmd, env, None
            started.append(self)

        def poll(self):
            self.returncode = 3 if self.env['RANK'] == '1' else 0
            return self.returncode

        def kill(self):
            pass
    import subprocess as sp
    monkeypatch.setattr(sp, 'Popen', FakeProc)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '2'])
    with pytest.raises(SystemExit) as ex:
        bench._self_launch()
    assert ex.value.code == 3 and len(started) == 4
    assert [p.env['RANK'] for p in started] == ['0', '1', '2', '3'] and [p.env['LOCAL_RANK'] for p in started] == ['0', '1', '2', '3']
    assert all(p.env['WORLD_SIZE'] == '4' and p.env['MASTER_ADDR'] == '127.0.0.1' and p.env['MASTER_PORT'].isdigit() for p in started)
    assert all(p.cmd[0] == sys.executable and p.cmd[1].endswith('bench.py') and p.cmd[2:] == ['--gpus', '4', '--steps', '2'] for p in started)
    del started[:]
    monkeypatch.setenv('WORLD_SIZE', '4')                   # under torchrun: this process IS a rank
    assert bench._self_launch() is None and not started
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.setattr(sys, 'argv', ['bench.py'])           # the default N = 1 run stays in-process
    assert bench._self_launch() is None and not started
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '1', '--spawn'])
    with pytest.raises(SystemExit):
        bench._self_launch()
    assert len(started) == 1 and started[0].env['WORLD_SIZE'] == '1'


def test_eight_rank_rehearsal_of_the_bench_launch_on_the_cpu(tmp_path):
    """`python bench.py --gpus 8 --rehearse` as the driver would type the 8-GPU command -- no launcher environment,
    bench.py starts the eight ranks itself -- with every rank on the CPU over gloo (eight processes may not share one GPU on this
    pool, and an 8-GPU node is not ours to launch on): port and environment propagation, all ranks joining, the broadcast making the
    replicas equal, the flat all-reduce with its failure-flag slot, the MAX-over-ranks timing and exactly ONE JSON line from rank 0."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'CRFCONV_DIST_BACKEND')}
    env['OMP_NUM_THREADS'] = '1'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--rehearse', '--steps', '3', '--warmup', '1', '--points', '4096']
    with open(str(tmp_path / 'out'), 'w') as fo, open(str(tmp_path / 'err'), 'w') as fe:
        rc = subprocess.run(cmd, env=env, stdout=fo, stderr=fe, cwd=ROOT, timeout=300).returncode
    assert rc == 0, open(str(tmp_path / 'err')).read()[-3000:]
    lines = [l for l in open(str(tmp_path / 'out')).read().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['rehearsal'] is True and rec['world'] == 8 and rec['dist_backend'] == 'gloo' and rec['n_gpus'] == 0
    assert rec['replica_spread'] == 0.0 and rec['guard_slot'] == 0.0 and rec['ms_per_step'] > 0
    # a rank that fails must surface as the parent's exit status (and the others must not be left behind)
    bad = subprocess.run(cmd + ['--points', '-5'], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT, timeout=300)
    assert bad.returncode != 0


def test_tolerances_are_tied_to_the_recorded_errors(monkeypatch):
    """tests/gpu_util.assert_close: with an entry in the baseline (key = test :: what, no occurrence counter) the enforced bound is
    min(stated, max(10 x recorded, noise floor)); the floor is 16 eps where the recorded run was bit-exact; a missing key warns (and
    fails under CRFCONV_TOL_STRICT)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import gpu_util
    monkeypatch.delenv('CRFCONV_TOL_RECORD', raising=False)
    monkeypatch.delenv('CRFCONV_TOL_STRICT', raising=False)
    gpu_util.set_current_test('tests/x.py::t')
    monkeypatch.setattr(gpu_util, '_baseline', {'tests/x.py::t::a': 1e-6, 'tests/x.py::t::z': 0.0, 'tests/x.py::t::b': 1e-3})
    monkeypatch.setattr(gpu_util, '_warned', set())
    ref = torch.ones(8)
    gpu_util.assert_close(ref + 5e-6, ref, 1e-4, 'a')        # 5e-6 <= 10 x 1e-6
    gpu_util.assert_close(ref + 5e-6, ref, 1e-4, 'a')        # a second call of the same site has the same key, the same bound
    with pytest.raises(AssertionError, match='recorded on MI355X'):
        gpu_util.assert_close(ref + 5e-5, ref, 1e-4, 'a')
    gpu_util.assert_close(ref + 1e-6, ref, 1e-4, 'z')        # recorded 0: floor 16 eps = 1.9e-6, not 1e-7
    with pytest.raises(AssertionError, match='recorded on MI355X'):
        gpu_util.assert_close(ref + 5e-6, ref, 1e-4, 'z')
    gpu_util.assert_close(ref + 5e-5, ref, 1e-4, 'b')        # recorded error above the stated bound: the stated bound rules
    with pytest.raises(AssertionError):
        gpu_util.assert_close(ref + 5e-4, ref, 1e-4, 'b')
    with pytest.warns(UserWarning, match='no recorded error'):
        gpu_util.assert_close(ref + 5e-5, ref, 1e-4, 'unrecorded site')
    monkeypatch.setenv('CRFCONV_TOL_STRICT', '1')
    with pytest.raises(AssertionError, match='CRFCONV_TOL_STRICT'):
        gpu_util.assert_close(ref + 5e-5, ref, 1e-4, 'another unrecorded site')
    baseline = __import__('json').load(open(gpu_util._BASELINE_PATH))
    assert len(baseline) > 2000 and all(v >= 0 for v in baseline.values()) and not any(k.rsplit('#', 1)[-1].isdigit() and '#' in k.rsplit('::', 1)[-1][-4:] for k in baseline)


def test_dropout_mask_host_twin_is_a_keyed_bernoulli_stream():
    """ops.dropout_keep_mask -- the host twin of csrc/common.hpp::dropout_keep (two 32-bit keys per (seed, counter) from one
    splitmix64 round, a two-multiply 32-bit finalizer per element): deterministic, keep rate 1 - p, another mask for another
    counter or seed, no correlation between neighbouring elements or rows, element numbering beyond 2^32 defined."""
    import numpy as np
    from crfconv_amd import ops
    n = 1 << 18
    m = ops.dropout_keep_mask(1234, 7, n, 0.5)
    assert m.dtype == np.bool_ and m.shape == (n,)
    assert np.array_equal(m, ops.dropout_keep_mask(1234, 7, n, 0.5))
    assert abs(m.mean() - 0.5) < 5e-3
    assert abs(ops.dropout_keep_mask(1234, 7, n, 0.3).mean() - 0.7) < 5e-3
    assert ops.dropout_keep_mask(1234, 7, n, 0.0).all()
    for other in (ops.dropout_keep_mask(1234, 8, n, 0.5), ops.dropout_keep_mask(1235, 7, n, 0.5)):
        assert abs((m != other).mean() - 0.5) < 5e-3                # independent fair coins differ half of the time
    for lag in (1, 2, 16, 128):
        c = np.corrcoef(m[:-lag].astype(np.float64), m[lag:].astype(np.float64))[0, 1]
        assert abs(c) < 1e-2, (lag, c)
    # a mask is a function of (seed, counter, element) only: a longer call starts with the shorter one
    assert np.array_equal(ops.dropout_keep_mask(99, 1, 1000, 0.5), ops.dropout_keep_mask(99, 1, 4000, 0.5)[:1000])


def test_graphed_model_state_dict_is_symmetric_as_a_submodule():
    """GraphedModel is transparent for checkpoints -- keys are the wrapped model's -- also when it is a submodule of another module:
    the parent's state_dict() and load_state_dict(strict=True) must agree on the keys (overriding the two methods covered the
    top-level call only)."""
    import torch
    from crfconv_amd.train import GraphedModel

    class Inner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin, self.bn = torch.nn.Linear(3, 2), torch.nn.BatchNorm1d(2)

    class Outer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net, self.head = GraphedModel(Inner()), torch.nn.Linear(2, 2)

    inner = Inner()
    g = GraphedModel(inner)
    assert list(g.state_dict().keys()) == list(inner.state_dict().keys())
    g.load_state_dict(inner.state_dict(), strict=True)
    a, b = Outer(), Outer()
    sd = a.state_dict()
    assert 'net.lin.weight' in sd and not any('.model.' in k for k in sd)
    b.load_state_dict(sd, strict=True)
    assert torch.equal(b.net.model.lin.weight, a.net.model.lin.weight) and torch.equal(b.net.model.bn.running_var, a.net.model.bn.running_var)
