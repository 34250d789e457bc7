"""-m gpu: the HIP model path (through the C ABI) against the CPU oracle and the fixtures captured
from the reference.  Tolerances: outputs / logits 1e-4 (BASELINE.json north_star), gradients 2e-4
of the tensor's largest magnitude (fp32, different but fixed summation orders)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import _seeded as S
from conftest import sub
from gpu_util import DEV, assert_close, assert_close_anchored, grads, load_sd, relerr, t
from oracle import crf_oracle as O
from oracle import native as onative

pytestmark = pytest.mark.gpu


@pytest.fixture
def big_forms_from_4096(monkeypatch):
    """Kernel-level tests of the row-streaming (big-level) forms use 4 100 ... 10 240 rows to stay quick; the shipped switch-over
    between the small and the big forms is at 12 288 rows (ops.state.mfma_min_rows, swept on the training step), so they pin it."""
    from crfconv_amd import ops
    monkeypatch.setattr(ops.state, 'mfma_min_rows', 4096)

OUT_TOL = 1e-4
GRAD_TOL = 2e-4


def _report_err(what, got, ref64):
    """max |got - ref| per element (absolute) beside the normalised figure the tolerances are stated on."""
    a = float((got.detach().cpu().double() - ref64).abs().max())
    n = a / max(1.0, float(ref64.abs().max()))
    print('%-40s max abs err %.3e   normalised %.3e   (max |ref| %.3e)' % (what, a, n, float(ref64.abs().max())))
    return a, n


def knn_tables(B, N, K, seed):
    pos = np.stack([S.make_cloud(seed + b, N) for b in range(B)])
    return pos, onative.oracle_knn_batch(pos, pos, K)


# ------------------------------------------------------------------ low-level ops
@pytest.mark.parametrize('C', [4, 32, 6, 512])
def test_gather_and_maxpool(C):
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    B, N, M, K = 2, 300, 77, 16
    rng = np.random.default_rng(C)
    idx = rng.integers(0, N, (B, M, K))
    x = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32))
    gout = torch.from_numpy(rng.standard_normal((B, M, C)).astype(np.float32))
    tab = NeighborTable(t(idx), N)
    xd = x.to(DEV).reshape(-1, C).requires_grad_(True)
    out = ops.neighbor_maxpool(xd, tab)
    out.backward(gout.to(DEV).reshape(-1, C))
    xr = x.clone().requires_grad_(True)
    ref = O._rows(xr, torch.from_numpy(idx)).max(2)[0]
    ref.backward(gout)
    assert_close(out.reshape(B, M, C), ref, 0, 'maxpool')
    assert_close(xd.grad.reshape(B, N, C), xr.grad, 1e-6, 'maxpool grad')

    up = rng.integers(0, N, (B, M, 1))
    tab1 = NeighborTable(t(up), N)
    xd = x.to(DEV).reshape(-1, C).requires_grad_(True)
    out = ops.gather_rows(xd, tab1)
    out.backward(gout.to(DEV).reshape(-1, C))
    xr = x.clone().requires_grad_(True)
    ref = O._rows(xr, torch.from_numpy(up))[:, :, 0]
    ref.backward(gout)
    assert_close(out.reshape(B, M, C), ref, 0, 'gather')
    assert_close(xd.grad.reshape(B, N, C), xr.grad, 1e-6, 'gather grad')


def test_table_rejects_bad_indices():
    from crfconv_amd.graph import NeighborTable
    idx = torch.zeros((1, 4, 3), dtype=torch.long)
    idx[0, 1, 2] = 9
    with pytest.raises(IndexError):
        NeighborTable(idx.to(DEV), 9)
    from crfconv_amd._lib import CrfConvError
    with pytest.raises(CrfConvError):
        NeighborTable(idx, 10)          # CPU tensor: no CPU path


def test_deferred_table_check_raises_without_a_host_sync_in_the_refresh():
    """batched_reverse(defer_check=True) -- MultiScaleData.load_(.., defer_check=True), the refresh in front of every
    train.GraphedModel replay: a refreshed table with out-of-range entries is reported by graph.check_pending() once its count has
    reached the host, not by a synchronising .item() per table inside the refresh; good tables pass and are marked checked."""
    from crfconv_amd.graph import NeighborTable, batched_reverse, check_pending, _PENDING_CHECKS
    good = torch.randint(0, 9, (1, 64, 4))
    bad = good.clone()
    bad[0, 5, 2] = 9
    bad[0, 7, 1] = -3
    ta, tb = NeighborTable(good.to(DEV), 9), NeighborTable(good.to(DEV), 9)
    with batched_reverse(defer_check=True):
        ta.refresh_(good.flip(1).to(DEV))
        tb.refresh_(bad.to(DEV))
    assert len(_PENDING_CHECKS) == 1 and not tb._checked
    with pytest.raises(IndexError, match='2 neighbour indices outside'):
        check_pending(wait=True)
    assert ta._checked and not _PENDING_CHECKS and int(tb._bad) == 0
    check_pending(wait=True)                                  # nothing pending: no error
    with batched_reverse(defer_check=True):
        tb.refresh_(good.to(DEV))
    check_pending(wait=True)
    assert tb._checked


def test_tables_built_inside_a_forward_are_checked_without_a_host_sync():
    """graph.table_of (what every layer calls on the batch's index tensors) builds its tables with the range check DEFERRED: no .item()
    per table -- thirteen of them per fresh batch serialised an eager step --; a bad table is reported by check_pending() once its count
    has reached the host (and by the next table_of / load_ that polls it).  NeighborTable(...) called directly still raises at once."""
    from crfconv_amd.graph import check_pending, table_of, _PENDING_CHECKS
    check_pending(wait=True)
    good = torch.randint(0, 9, (1, 64, 4)).to(DEV)
    bad = good.clone()
    bad[0, 3, 1] = 11
    tg = table_of(good, 9)
    tb = table_of(bad, 9)                                      # no error here: the count is still on its way
    assert not tb._checked and len(_PENDING_CHECKS) >= 1
    with pytest.raises(IndexError, match='1 neighbour indices outside'):
        check_pending(wait=True)
    assert tg._checked and not _PENDING_CHECKS
    assert int(tb.idx32.max()) <= 8                            # the entry was clamped: nothing could fault meanwhile


@pytest.mark.parametrize('K', [1, 5, 16, 32, 40])
def test_table_columns_sorted_keep_row_content(K):
    """The device table re-orders columns 1.. of a row by ascending source id (locality of the gathers): same
    multiset per row, column 0 untouched, uint16 local ids consistent with the int32 global rows."""
    from crfconv_amd.graph import NeighborTable
    B, n_tgt, n_src = 3, 777, 1500
    rng = np.random.default_rng(K)
    idx = rng.integers(0, n_src, (B, n_tgt, K))
    tab = NeighborTable(t(idx), n_src)
    got = tab.idx32.cpu().numpy().reshape(B, n_tgt, K) - (np.arange(B) * n_src)[:, None, None]
    assert np.array_equal(got[:, :, 0], idx[:, :, 0])
    assert np.array_equal(np.sort(got[:, :, 1:], -1), np.sort(idx[:, :, 1:], -1))
    assert np.all(np.diff(got[:, :, 1:], axis=-1) >= 0)
    if tab.idx16 is not None:
        assert np.array_equal(tab.idx16.cpu().numpy().astype(np.int64).reshape(B, n_tgt, K) & 0xffff, got)


@pytest.mark.parametrize('H,K,steps,B', [(8, 16, 3, 2), (16, 16, 1, 1), (32, 32, 5, 2), (64, 16, 3, 2),
                                         (8, 16, 0, 2), (12, 9, 2, 3), (4, 16, 3, 2)])
def test_meanfield_vs_oracle(H, K, steps, B):
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    N = 700
    pos, nbr = knn_tables(B, N, K, 40 + H)
    z = S.uniform(H, 'z', (B, N, H))
    y = S.uniform(H, 'y', (B, N, H))
    c = (np.eye(H) + 0.1 * S.uniform(H, 'c', (H, H))).astype(np.float32)
    g = S.uniform(H, 'g', (B, N, H))
    # oracle
    zr, yr, cr = (torch.from_numpy(a).requires_grad_(True) for a in (z, y, c))
    ref = O.crf_meanfield(zr, yr, torch.from_numpy(nbr)[:, :, 1:], cr, steps)
    (ref * torch.from_numpy(g)).sum().backward()
    # HIP
    tab = NeighborTable(t(nbr), N)
    zd, yd, cd = (t(a).requires_grad_(True) for a in (z, y, c))
    out = ops.crf_meanfield(zd.reshape(-1, H), yd.reshape(-1, H), cd, tab, steps, k0=1)
    (out.reshape(B, N, H) * t(g)).sum().backward()
    assert_close(out.reshape(B, N, H), ref, OUT_TOL, 'x_T')
    zero = lambda g, like: torch.zeros_like(like) if g is None else g      # steps == 0: y, c unused
    assert_close(zd.grad, zr.grad, GRAD_TOL, 'dz')
    assert_close(zero(yd.grad, yd), zero(yr.grad, yr), GRAD_TOL, 'dy')
    assert_close(zero(cd.grad, cd), zero(cr.grad, cr), GRAD_TOL, 'dc')


@pytest.mark.parametrize('H,K,steps', [(8, 16, 3), (16, 16, 3), (32, 16, 2), (64, 16, 2), (8, 32, 2), (8, 16, 1)])
def test_meanfield_backward_hub_rows_and_rows_without_in_edges(H, K, steps):
    """Reverse walks over very uneven in-degrees: three columns of every row point at rows 5, 6 and N - 1 of its cloud (in-degrees of
    ~N: a wavefront's range of the reverse edge list spans many chunks, rows continue across chunk boundaries, pair sums straddle
    them), the rest at a narrow band of rows, so that most rows have NO in-edge at all; N is odd and no multiple of the rows per
    wavefront.  Forward and every gradient against the float32 oracle (crf_oracle.crf_meanfield)."""
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    B, N = 2, 1237
    rng = np.random.default_rng(100 + H + K)
    nbr = rng.integers(40, 90, (B, N, K))                     # a band of 50 rows takes almost every edge
    nbr[:, :, 0] = np.arange(N)
    nbr[:, :, 3], nbr[:, :, 7], nbr[:, :, K - 1] = 5, 6, N - 1  # hubs
    nbr[:, ::7, 2] = rng.integers(0, N, (B, len(range(0, N, 7))))      # a few edges anywhere
    z = S.uniform(H, 'z', (B, N, H))
    y = S.uniform(H, 'y', (B, N, H))
    c = (np.eye(H) + 0.1 * S.uniform(H, 'c', (H, H))).astype(np.float32)
    g = S.uniform(H, 'g', (B, N, H))
    zr, yr, cr = (torch.from_numpy(a).requires_grad_(True) for a in (z, y, c))
    ref = O.crf_meanfield(zr, yr, torch.from_numpy(nbr)[:, :, 1:], cr, steps)
    (ref * torch.from_numpy(g)).sum().backward()
    tab = NeighborTable(t(nbr), N)
    zd, yd, cd = (t(a).requires_grad_(True) for a in (z, y, c))
    out = ops.crf_meanfield(zd.reshape(-1, H), yd.reshape(-1, H), cd, tab, steps, k0=1)
    (out.reshape(B, N, H) * t(g)).sum().backward()
    assert_close(out.reshape(B, N, H), ref, OUT_TOL, 'hub table: x_T')
    # (hub rows sum ~N terms: the comparison is relative to the largest gradient entry, as everywhere)
    assert_close(zd.grad, zr.grad, GRAD_TOL, 'hub table: dz')
    assert_close(yd.grad, yr.grad, GRAD_TOL, 'hub table: dy')
    assert_close(cd.grad, cr.grad, GRAD_TOL, 'hub table: dc')


def _local_table(B, N, K, seed, spread):
    """[B, N, K] int64 neighbours: column 0 = self, the others within +-spread rows of the target (a spatially sorted cloud in
    miniature) or anywhere (spread = 0: a shuffled cloud)."""
    g = torch.Generator().manual_seed(seed)
    i = torch.arange(N).reshape(1, N, 1)
    if spread:
        j = (i + torch.randint(-spread, spread + 1, (B, N, K), generator=g)).clamp_(0, N - 1)
    else:
        j = torch.randint(0, N, (B, N, K), generator=g)
    j[:, :, 0] = torch.arange(N)
    return j


@pytest.mark.parametrize('B,N,spread,steps,big', [(2, 5000, 300, 3, False), (1, 3000, 0, 3, False), (3, 4099, 150, 1, False), (2, 700, 40, 5, False),
                                                   (4, 40960, 400, 3, False), (2, 70001, 500, 2, True), (2, 40960, 0, 2, False), (2, 40960, 2000, 3, False)])
def test_block_resident_meanfield_forward_equals_the_per_step_launches(B, N, spread, steps, big):
    """csrc/crf_block.hip (one launch, block-resident rows, grid barriers between the steps) against csrc/crf.hip's per-step launches:
    weights s and x_1 bit for bit, the later iterates to rounding (their messages are added in-block columns first) -- local tables (most neighbours served from LDS), a shuffled one (every neighbour from
    the row tables past L1; at 2 x 40 960 points more references per block than the LDS halo holds: the overflow path), ragged last blocks, one to five steps, uint16 and int32 (clouds of more than 65 536 points) index rows;
    the barrier words are left zero, no failure code; outputs and gradients through ops.crf_meanfield agree to rounding."""
    from crfconv_amd import _lib, ops
    from crfconv_amd.graph import NeighborTable, ptr, stream_ptr
    from crfconv_amd.ops._base import gridsync_ws
    H, K = 8, 16
    m = B * N
    tab = NeighborTable(_local_table(B, N, K, N + steps, spread).to(DEV), N)
    assert (tab.idx16 is None) == big
    g = torch.Generator().manual_seed(N)
    z = torch.randn(m, H, generator=g).to(DEV)
    y = (0.7 * torch.randn(m, H, generator=g)).to(DEV)
    c = torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)
    C = c.t() @ c
    Q = torch.linalg.inv(torch.eye(H) + C)
    P = (C @ Q).to(DEV).contiguous()
    Q = Q.to(DEV).contiguous()
    rows = _lib.load().crfconv_meanfield_forward_block_rows(m, H, K, 1, steps)
    assert rows > 0 and rows % 64 == 0
    ws = gridsync_ws(torch.device(DEV))
    res = []
    for form in ('steps', 'block', 'block'):           # the block form twice: the barrier words of the first launch serve the second
        s_ = torch.full((m, K), float('nan'), device=DEV)
        xs = torch.full((steps, m, H), float('nan'), device=DEV)
        if form == 'block':
            _lib.call('crfconv_meanfield_forward_block', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, K, 1, m, H,
                      ptr(Q), ptr(P), steps, ptr(s_), ptr(xs), ptr(ws), stream_ptr())
        else:
            _lib.call('crfconv_meanfield_forward_u16', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, K, 1, m, H,
                      ptr(Q), ptr(P), steps, ptr(s_), ptr(xs), stream_ptr())
        torch.cuda.synchronize()
        res.append((s_, xs))
    assert int(ws.abs().sum()) == 0, 'barrier words not left zero (failure word: %d)' % int(ws[_lib.load().crfconv_gridsync_fail_word()])
    for s_, xs in res[1:]:
        assert bool(torch.isfinite(xs).all()) and torch.equal(s_, res[0][0]) and torch.equal(xs[0], res[0][1][0])
        for t_ in range(1, steps):           # later steps add their message in-block columns first: equal to rounding
            assert_close(xs[t_], res[0][1][t_], 2e-6, 'block-resident x_%d vs per-step launches' % (t_ + 1))
    assert torch.equal(res[1][1], res[2][1])           # and reproducible
    # inference with one step: no weight store
    if steps == 1:
        xs = torch.empty((1, m, H), device=DEV)
        _lib.call('crfconv_meanfield_forward_block', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, K, 1, m, H,
                  ptr(Q), ptr(P), 1, None, ptr(xs), ptr(ws), stream_ptr())
        assert torch.equal(xs, res[0][1])
    # through the operator: 'on' forces the block form, 'off' the per-step launches; outputs and gradients agree bit for bit
    outs = {}
    for mode in ('off', 'on'):
        ops.state.mf_block = mode
        try:
            zd, yd, cd = z.clone().requires_grad_(True), y.clone().requires_grad_(True), c.to(DEV).requires_grad_(True)
            out = ops.crf_meanfield(zd, yd, cd, tab, steps)
            (out * torch.linspace(-1, 1, out.numel(), device=DEV).reshape(out.shape)).sum().backward()
            outs[mode] = (out.detach(), zd.grad, yd.grad, cd.grad)
        finally:
            ops.state.mf_block = 'auto'
    for a, b, what in zip(outs['off'], outs['on'], ('x_T', 'dz', 'dy', 'dc')):       # ('on': the backward's edge pass in block form too)
        assert_close(b, a, 5e-6, 'block-resident mean field vs per-step launches: ' + what)


def test_block_resident_meanfield_is_chosen_for_local_tables_only():
    """ops.crf._block_rows ('auto'): a table whose entries stay inside the target's block of rows takes the one-launch form, a shuffled
    one and a small one do not; the measured fraction is cached on the table."""
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    from crfconv_amd.ops.crf import _block_rows
    B, N = 2, 40960
    near = NeighborTable(_local_table(B, N, 16, 1, 100).to(DEV), N)
    far = NeighborTable(_local_table(B, N, 16, 2, 0).to(DEV), N)
    small = NeighborTable(_local_table(1, 4096, 16, 3, 50).to(DEV), 4096)
    assert ops.state.mf_block == 'auto'
    # the first question about a table leaves the measurement in flight and answers 'per-step launches' (no host synchronisation); the second reads it
    from crfconv_amd import _lib
    rows = _lib.load().crfconv_meanfield_forward_block_rows(B * N, 8, 16, 1, 3)
    assert rows > 0
    assert _block_rows(near, B * N, 8, 1, 3) == 0 and ('block_locality_pending', rows) in near.cache
    assert _block_rows(near, B * N, 8, 1, 3) == rows and near.cache[('block_locality', rows)] > 0.7 and ('block_locality_pending', rows) not in near.cache
    assert _block_rows(far, B * N, 8, 1, 3) == 0 and _block_rows(far, B * N, 8, 1, 3) == 0 and far.cache[('block_locality', rows)] < 0.05
    assert _block_rows(small, 4096, 8, 1, 3) == 0 and _block_rows(near, B * N, 16, 1, 3) == 0 and _block_rows(near, B * N, 8, 1, 0) == 0


def test_meanfield_fp64_anchor(golden):
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    g = golden('g2_meanfield_fp64.npz')
    B, N, H = g['z'].shape
    nbr = np.concatenate([np.zeros((B, N, 1), np.int64), g['nbr'].astype(np.int64)], -1)   # dummy self column
    tab = NeighborTable(t(nbr), N)
    for steps in (1, 3, 5):
        out = ops.crf_meanfield(t(g['z'], torch.float32).reshape(-1, H), t(g['y'], torch.float32).reshape(-1, H),
                                t(g['c'], torch.float32), tab, steps)
        assert_close(out.reshape(B, N, H), g['x_T%d' % steps], 2e-5, 'x_T%d vs fp64' % steps)


# ------------------------------------------------------------------ modules vs reference fixtures
@pytest.mark.parametrize('steps,mode', [(1, 'train'), (3, 'train'), (3, 'eval'), (5, 'eval')])
def test_crfconv_module_golden(golden, steps, mode):
    from crfconv_amd.models import ContinuousGaussianCRFConv
    g = golden('g1_crfconv.npz')
    m = load_sd(ContinuousGaussianCRFConv(64, 32, 32, steps=steps), sub(g, 'sd')).to(DEV)
    m.train(mode == 'train')
    u = t(g['unary']).requires_grad_(True)
    p = t(g['pairwise']).requires_grad_(True)
    out = m(u, p, t(g['up_idx'], torch.long), t(g['neighbor_idx'], torch.long))
    (out * t(g['gout'])).sum().backward()
    tag = 'T%d_%s' % (steps, mode)
    assert_close(out, g[tag + '/out'], OUT_TOL, 'out')
    assert_close(u.grad, g[tag + '/d_unary'], GRAD_TOL, 'd_unary')
    assert_close(p.grad, g[tag + '/d_pairwise'], GRAD_TOL, 'd_pairwise')
    gr = grads(m)
    for k, v in sub(g, tag + '/grad').items():
        assert_close(gr[k], v, GRAD_TOL, 'grad ' + k)
    if mode == 'train':
        sd = m.state_dict()
        for k, v in sub(g, tag + '/buf').items():
            if 'num_batches' in k:
                assert int(sd[k]) == int(v), k
            else:
                assert_close(sd[k], v, 1e-5, k)


@pytest.mark.parametrize('form', ['plain', 'strided'])
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_pointconv_module_golden(golden, form, mode):
    from crfconv_amd.models import PointConv
    g = golden('g3_pointconv.npz')
    m = load_sd(PointConv(8), sub(g, 'sd')).to(DEV)
    m.train(mode == 'train')
    x = t(g['x']).requires_grad_(True)
    if form == 'plain':
        out = m(x, t(g['pos']), t(g['neighbor_idx'], torch.long))
    else:
        out = m(x, (t(g['pos']), t(g['sub_pos'])), t(g['sub_idx'], torch.long))
    (out * t(g[form + '/gout'])).sum().backward()
    tag = '%s_%s' % (form, mode)
    assert_close(out, g[tag + '/out'], OUT_TOL, 'out')
    assert_close(x.grad, g[tag + '/d_x'], GRAD_TOL, 'd_x')
    gr = grads(m)
    for k, v in sub(g, tag + '/grad').items():
        assert_close(gr[k], v, GRAD_TOL, 'grad ' + k)
    if mode == 'train':
        sd = m.state_dict()
        for k, v in sub(g, tag + '/buf').items():
            if 'num_batches' in k:
                assert int(sd[k]) == int(v), k
            else:
                assert_close(sd[k], v, 1e-5, k)


@pytest.mark.parametrize('d,deferred', [(4, False), (16, False), (32, False), (64, False), (128, False), (16, True), (32, True),
                                        (64, True), (128, True)])
def test_pointconv_widths_vs_oracle(d, deferred):
    """PointConv forward, input gradient and every parameter gradient against the float32 oracle anchored on float64, all widths.
    deferred: the backward under ops.deferred_weight_grads -- the wide layers' parameter pass then runs at the end of the pass, at
    d = 32 / 64 as ONE matrix-pipe launch without per-edge tensors (csrc/pointconv_wide.hip), at d = 128 as dump + GEMM passes."""
    from crfconv_amd import ops
    from crfconv_amd.models import PointConv
    B, N, K = 2, 200, 16
    pos, nbr = knn_tables(B, N, K, 60 + d)
    m = PointConv(d)
    sd = S.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 31)
    m.load_state_dict(sd)
    x = S.uniform(d, 'x', (B, N, d))
    gout = S.uniform(d, 'g', (B, N, d))
    def oracle(dtype):
        prm = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(
            v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
        xr = torch.from_numpy(x).to(dtype).requires_grad_(True)
        ref = O.point_conv(prm, '', xr, torch.from_numpy(pos).to(dtype), torch.from_numpy(nbr), True)
        (ref * torch.from_numpy(gout).to(dtype)).sum().backward()
        return prm, xr, ref
    prm, xr, ref = oracle(torch.float32)
    prm64, xr64, ref64 = oracle(torch.float64)
    m = m.to(DEV).train()
    xd = t(x).requires_grad_(True)
    out = m(xd, t(pos), t(nbr))
    with ops.deferred_weight_grads(enabled=deferred):
        (out * t(gout)).sum().backward()
    assert_close(out, ref, OUT_TOL, 'out')
    assert_close(xd.grad, xr.grad, GRAD_TOL, 'd_x')
    # LeakyReLU kink: a channel whose layer-1 pre-activation sits within fp32 rounding of 0 on some
    # edge (all self edges share rel = 0, so this happens for whole groups) has a one-sided derivative
    # that ANY fp32 evaluation order may resolve either way; such channels are compared on outputs only.
    P64 = torch.from_numpy(pos).double()
    rel = (P64.unsqueeze(2) - O._rows(P64, torch.from_numpy(nbr))).reshape(-1, 3)
    h = rel @ sd['weight_nn.0.lin.weight'].double().t()
    pre = (h - h.mean(0)) / torch.sqrt(h.var(0, unbiased=False) + 1e-5) * sd['weight_nn.0.bn.batch_norm.weight'].double() \
        + sd['weight_nn.0.bn.batch_norm.bias'].double()
    ok = (pre.abs().min(0).values > 1e-5).to(DEV)
    for k, v in grads(m).items():
        a, b, c = v, prm[k].grad.to(DEV), prm64[k].grad.to(DEV)
        if k.startswith('weight_nn.0.'):
            a, b, c = a[ok], b[ok], c[ok]
        assert_close_anchored(a, b, c, GRAD_TOL, 'grad ' + k)


@pytest.mark.parametrize('d', [32, 64, 128])
@pytest.mark.parametrize('deferred', [False, True])
def test_pointconv_wide_layers_on_a_padded_table_with_missing_edges(d, deferred):
    """The matrix-pipe kernels of the wide layers (csrc/pointconv_wide.hip: forward statistics pass at d >= 32, parameter pass at d = 32 / 64
    under deferred weight gradients) take one target point = sixteen table entries per MFMA tile; entries < 0 (a padded, variable-degree
    table: graph.table_from_edges) must contribute nothing.  Ragged in-degrees 3 .. 16, a target count that no workgroup partition divides,
    separate source / target point sets; checked against a float64 restatement of models/point_conv_big.py:37-58 on the edge list."""
    from crfconv_amd import ops
    from crfconv_amd.graph import table_from_edges
    g = torch.Generator().manual_seed(100 + d)
    n_tgt, n_src = 333, 517
    deg = torch.randint(3, 17, (n_tgt,), generator=g)
    deg[7] = 16
    tgt = torch.repeat_interleave(torch.arange(n_tgt), deg)
    src = torch.cat([torch.randperm(n_src, generator=g)[:int(k)] for k in deg])
    pos_s, pos_t = torch.rand(n_src, 3, generator=g), torch.rand(n_tgt, 3, generator=g)
    x0 = torch.randn(n_src, d, generator=g)
    gout = torch.randn(n_tgt, d, generator=g)
    W1_0, W2_0 = torch.randn(d, 3, generator=g), torch.randn(d, d, generator=g) / d ** 0.5
    gam1, bet1, gam2, bet2 = (torch.rand(d, generator=g) + 0.5 for _ in range(4))

    def reference():
        x, W1, W2, g1, b1, g2, b2 = (v.double().clone().requires_grad_(True) for v in (x0, W1_0, W2_0, gam1, bet1, gam2, bet2))

        def bn(h, gm, bt):
            return (h - h.mean(0)) / torch.sqrt(h.var(0, unbiased=False) + 1e-5) * gm + bt
        rel = pos_t.double()[tgt] - pos_s.double()[src]
        w = bn(torch.nn.functional.leaky_relu(bn(rel @ W1.t(), g1, b1), 0.1) @ W2.t(), g2, b2)
        out = torch.zeros(n_tgt, d, dtype=torch.float64).index_add(0, tgt, w * x[src])
        (out * gout.double()).sum().backward()
        return out.detach(), [v.grad for v in (x, W1, W2, g1, b1, g2, b2)]
    ref_out, ref_grads = reference()
    table = table_from_edges(tgt.to(DEV), src.to(DEV), n_tgt, n_src)
    assert table.K == 16 and bool((table.idx32 < 0).any())
    x = x0.to(DEV).requires_grad_(True)
    W1, W2 = nn.Parameter(W1_0.to(DEV)), nn.Parameter(W2_0.to(DEV))
    bn1, bn2 = nn.BatchNorm1d(d).to(DEV).train(), nn.BatchNorm1d(d).to(DEV).train()
    with torch.no_grad():
        bn1.weight.copy_(gam1); bn1.bias.copy_(bet1); bn2.weight.copy_(gam2); bn2.bias.copy_(bet2)
    out = ops.point_conv(x, pos_s.to(DEV), pos_t.to(DEV), table, W1, bn1, W2, bn2, True)
    with ops.deferred_weight_grads(enabled=deferred):
        (out * gout.to(DEV)).sum().backward()
    assert_close(out, ref_out, OUT_TOL, 'out')
    for got, want, name in zip((x.grad, W1.grad, W2.grad, bn1.weight.grad, bn1.bias.grad, bn2.weight.grad, bn2.bias.grad), ref_grads,
                               ('dx', 'dW1', 'dW2', 'dgamma1', 'dbeta1', 'dgamma2', 'dbeta2')):
        assert_close(got, want, GRAD_TOL, name)


@pytest.mark.parametrize('name,mode', [('a', 'train'), ('a', 'eval'), ('b', 'train'), ('c', 'train'), ('c', 'eval')])
def test_resblock_module_golden(golden, name, mode):
    from crfconv_amd.models import ResNetBBlock
    g = golden('g4_resblock.npz')
    cin, cout = {'a': (6, 32), 'b': (32, 32), 'c': (32, 64)}[name]
    m = load_sd(ResNetBBlock(cin, cout), sub(g, name + '/sd')).to(DEV)
    m.train(mode == 'train')
    x = t(g[name + '/x']).requires_grad_(True)
    if name == 'c':
        out = m(x, (t(g['pos']), t(g['sub_pos'])), t(g['sub_idx'], torch.long))
    else:
        out = m(x, t(g['pos']), t(g['neighbor_idx'], torch.long))
    (out * t(g[name + '/gout'])).sum().backward()
    tag = '%s_%s' % (name, mode)
    assert_close(out, g[tag + '/out'], OUT_TOL, 'out')
    assert_close(x.grad, g[tag + '/d_x'], GRAD_TOL, 'd_x')
    gr = grads(m)
    for k, v in sub(g, tag + '/grad').items():
        assert_close(gr[k], v, GRAD_TOL, 'grad ' + k)


class FixedDropout(nn.Module):
    """Dropout(0.5) with the mask the reference drew (captured in the fixture)."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask * 2.0 if self.training else x


@pytest.mark.parametrize('use_crf', [True, False])
def test_pointconvbig_golden(golden, use_crf):
    """Whole network + training-step contract (trainval.py:99-105), B=2, N=4096, with the multiscale
    tables REBUILT by the HIP kNN (and asserted equal to the reference's)."""
    import crfconv_amd
    from crfconv_amd import models
    g = golden('g5_pointconvbig.npz')
    tagc = 'crf' if use_crf else 'ups'
    B, N = g['pos'].shape[:2]
    choices = []
    n = N
    for i in range(5):
        choices.append(torch.from_numpy(S.permutation(5, 'choice%d' % i, n)[: n // (4, 4, 4, 4, 2)[i]]))
        n //= (4, 4, 4, 4, 2)[i]
    data = crfconv_amd.multiscale_compute(t(g['pos']), x=t(g['feats']), choices=choices)
    for i in range(5):
        for k in ('neighbor_idx', 'sub_idx', 'up_idx'):
            assert np.array_equal(getattr(data.multiscale[i], k).cpu().numpy(), g['ms%d/%s' % (i, k)].astype(np.int64)), (i, k)
    shapes = {k: tuple(int(s) for s in sh.split(',')) if sh else () for k, sh in zip(g[tagc + '/keys'], g[tagc + '/shapes'])}
    net = models.PointConvBig(6, 13, use_crf=use_crf, steps=int(g['steps']))
    net.load_state_dict(S.fill_state_dict(shapes, 17), strict=True)
    net = net.to(DEV)
    rows = g['rows']
    net.eval()
    with torch.no_grad():
        lg = net(data)
    assert lg.shape == (B * N, 13)
    assert_close(lg[rows], g[tagc + '_eval/logits_rows'], OUT_TOL, 'eval logits')
    assert_close(lg.double().sum(0), g[tagc + '_eval/logits_colsum'], OUT_TOL, 'eval colsum')

    mask = np.unpackbits(g[tagc + '_train/dropout_mask'])[: B * N * 128].reshape(B, N, 128)
    net.classifier[1] = FixedDropout(t(mask).float())
    net.train()
    logits = net(data)
    y = t(g['labels'], torch.long).reshape(-1) - 1
    loss = torch.nn.functional.cross_entropy(logits, y, weight=t(g['class_weights']), ignore_index=-1)
    loss.backward()
    # the stated bar (north_star: 1e-4).  Measured on MI355X, round 3 (CRFCONV_TEST_REPORT=1): 1.3e-6 -- rounds 1-2 allowed
    # 5e-4 here (train-mode BatchNorm over as few as 32 rows at level 4) without ever needing it
    assert_close(logits[rows], g[tagc + '_train/logits_rows'], 1e-4, 'train logits')
    assert_close(loss, g[tagc + '_train/loss'], 1e-4, 'loss')
    gr = grads(net)
    for k, v in sub(g, tagc + '_train/gnorm').items():
        got = gr[k].double().reshape(-1)
        nrm = float(v)
        scale = max(nrm, 1e-3)
        assert abs(float(got.norm()) - nrm) <= 2e-3 * scale, k
        proj = torch.from_numpy(S.projections(17, k, got.numel())).double().to(DEV) @ got
        want = torch.from_numpy(g['%s_train/gproj/%s' % (tagc, k)]).to(DEV)
        assert float((proj - want).abs().max()) <= 2e-3 * scale * np.sqrt(got.numel()), k
    for k, v in sub(g, tagc + '_train/grad').items():
        assert_close(gr[k], v, 2e-4, 'grad ' + k)            # measured worst: 1.6e-5 (conv1_2.lin_in.lin.weight); was 2e-3


def test_b1_is_supported():
    """The reference crashes at B == 1 (squeeze bug, continuous_crf_conv_big.py:43); here B == 1
    must equal the first cloud of a B == 2 batch in eval mode."""
    import crfconv_amd
    from crfconv_amd import models
    pos = np.stack([S.make_cloud(70 + b, 4096, box=(2, 2, 1)) for b in range(2)])
    feats = np.concatenate([pos, S.uniform(70, 'rgb', (2, 4096, 3), 0, 1)], -1)
    gsub = torch.Generator().manual_seed(3)
    choices, n = [], 4096
    for r in (4, 4, 4, 4, 2):
        choices.append(torch.randperm(n, generator=gsub)[: n // r])
        n //= r
    net = models.PointConvBig(6, 13, True, 3)
    net.load_state_dict(S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 3))
    net = net.to(DEV).eval()
    with torch.no_grad():
        two = net(crfconv_amd.multiscale_compute(t(pos), x=t(feats), choices=choices))
        one = net(crfconv_amd.multiscale_compute(t(pos[:1]), x=t(feats[:1]), choices=choices))
    assert_close(one, two[:4096], OUT_TOL, 'B=1 vs B=2')


def test_collate_graph_replay_equals_eager_collate():
    """data.CollateGraph: the device collate + the in-place refresh of the static batch as ONE hipGraph replay.  After
    cg.run(new clouds) the static batch (tensors AND derived tables / reverse CSRs / moments) must equal an eager collate of
    the same clouds with the same subsets, for several successive batches through the same captured graph."""
    import crfconv_amd
    from crfconv_amd import models, ops
    from crfconv_amd.data import CollateGraph
    from crfconv_amd.graph import table_of
    B, N = 2, 4096

    def clouds(seed):
        pos = np.stack([S.make_cloud(seed + b, N, box=(2, 2, 1)) for b in range(B)])
        feats = np.concatenate([pos, S.uniform(seed, 'rgb', (B, N, 3), 0, 1)], -1)
        return t(pos), t(feats), t(S.integers(seed, 'y', (B, N), 0, 14))
    pos0, x0, y0 = clouds(500)
    static = crfconv_amd.multiscale_compute(pos0, x=x0, y=y0, generator=torch.Generator().manual_seed(1))
    net = models.PointConvBig(6, 13, True, 3).to(DEV).train()
    ops.training_loss(net(static), static.y, None, ignore_index=-1).backward()       # every table / CSR / moment exists
    cg = CollateGraph(static, generator=torch.Generator().manual_seed(7))
    for seed in (510, 520, 530):
        pos, x, y = clouds(seed)
        cg.run(pos, x, y)
        ref = crfconv_amd.multiscale_compute(pos, x=x, y=y, choices=[c.clone() for c in cg.choices], sort='morton')
        assert torch.equal(static.x, ref.x) and torch.equal(static.y, ref.y)
        for a, b in zip(static.multiscale, ref.multiscale):
            for name in ('pos', 'neighbor_idx', 'sub_idx', 'up_idx'):
                assert torch.equal(getattr(a, name), getattr(b, name)), (seed, name)
            for name in ('neighbor_idx', 'sub_idx', 'up_idx'):
                idx_s, idx_r = getattr(a, name), getattr(b, name)
                for key, (tab_s, _) in getattr(idx_s, '_crf_tables', {}).items():      # (the last level's sub / up tables are never built)
                    tab_r = table_of(idx_r, key[0])
                    assert torch.equal(tab_s.idx32, tab_r.idx32)
                    if tab_s._rev is not None:
                        assert all(torch.equal(u, v) for u, v in zip(tab_s.reverse, tab_r.reverse))
                    assert int(tab_s._bad.item()) == 0
    # subset seeds: a draw on the caller's generator per graph (graphs rebuilt from one generator must not repeat
    # the same subset sequence), reproducible from equal generator states, resumable through state_dict
    g7 = torch.Generator().manual_seed(7)
    a, b = CollateGraph(static, generator=g7), CollateGraph(static, generator=g7)
    assert a.seed != b.seed and a.seed == cg.seed
    sd = cg.state_dict()
    assert sd['seed'] == cg.seed and sd['counter'] == 3
    pos, x, y = clouds(540)
    cg.run(pos, x, y)
    want = [c.clone() for c in cg.choices]
    b.load_state_dict(sd)                                   # another graph continues the first one's sequence
    b.run(pos, x, y)
    assert all(torch.equal(u, v) for u, v in zip(want, b.choices))


@pytest.mark.parametrize('gate', [False, True])
def test_collate_pipeline_double_buffer_overlaps_without_races(gate):
    """data.CollatePipeline: the collate graph of batch i+1 runs on a side stream while the consumer of batch i runs on the
    caller's stream.  Six batches through two slots; what the consumer sees after acquire() must equal an eager collate of
    the clouds submitted for that slot with the subsets drawn for it -- also with a slow consumer (a long kernel queue on
    the caller's stream between acquire and release) and a slot that is overwritten right after release.  gate: the collate graphs
    start behind the bounded device-side wait (crfconv_gate_wait) for a mark the consumer's stream makes (pipe.mark()) -- every
    second batch here, so that both the opened gate and the timed-out one (no mark: the collate goes ahead after GATE_MAX_WAIT_US)
    are seen; the batches must be the same either way."""
    import crfconv_amd
    from crfconv_amd import models, ops
    from crfconv_amd.data import CollatePipeline
    from crfconv_amd.graph import table_of
    B, N = 2, 4096

    def clouds(seed):
        pos = np.stack([S.make_cloud(seed + b, N, box=(2, 2, 1)) for b in range(B)])
        feats = np.concatenate([pos, S.uniform(seed, 'rgb', (B, N, 3), 0, 1)], -1)
        return t(pos), t(feats), t(S.integers(seed, 'y', (B, N), 0, 14))
    net = models.PointConvBig(6, 13, True, 3).to(DEV).train()
    statics = []
    for k in range(2):
        pos0, x0, y0 = clouds(600 + k)
        st = crfconv_amd.multiscale_compute(pos0, x=x0, y=y0, generator=torch.Generator().manual_seed(k))
        ops.training_loss(net(st), st.y, None, ignore_index=-1).backward()
        statics.append(st)
    pipe = CollatePipeline(statics, generator=torch.Generator().manual_seed(3), gate=gate)
    inputs = [clouds(700 + 10 * i) for i in range(7)]
    pipe.submit(1, *inputs[0])                                      # (first use of a slot captures its graph: both before the gate is on)
    pipe.submit(0, *inputs[0])
    pipe.enable_gate(True)
    seen = []
    burn = torch.randn(2048, 2048, device=DEV)
    for i in range(6):
        s = i % 2
        pipe.submit(1 - s, *inputs[i + 1])
        if i % 2 == 0:
            pipe.mark()                                              # (the training step's mark: opens the gate of the collate just queued)
        batch = pipe.acquire(s)
        for _ in range(4 if i % 3 == 0 else 0):
            burn = torch.tanh(burn @ burn * 1e-3)                      # a slow consumer: the batch must stay intact under it
        snap = {'x': batch.x.clone(), 'choices': [c.clone() for c in pipe.graphs[s].choices],
                'lv': [{n: getattr(lv, n).clone() for n in ('pos', 'neighbor_idx', 'sub_idx', 'up_idx')} for lv in batch.multiscale],
                'tab': [[(key, tab.idx32.clone()) for key, (tab, _) in lv.neighbor_idx._crf_tables.items()]
                        for lv in batch.multiscale]}
        pipe.release(s)
        seen.append(snap)
    torch.cuda.synchronize()
    assert pipe.gate_timeouts() == (3 if gate else 0)                # the three collates without a mark went ahead on their own
    assert pipe.gate_is_on() == bool(gate)                           # (never three in a row: the gate stayed on)
    for i, snap in enumerate(seen):
        pos, x, y = inputs[i]
        ref = crfconv_amd.multiscale_compute(pos, x=x, y=y, choices=snap['choices'], sort='morton')
        assert torch.equal(snap['x'], ref.x), i
        for lv_s, lv_r, tabs in zip(snap['lv'], ref.multiscale, snap['tab']):
            for n in ('pos', 'neighbor_idx', 'sub_idx', 'up_idx'):
                assert torch.equal(lv_s[n], getattr(lv_r, n)), (i, n)
            assert tabs
            for key, idx32 in tabs:
                assert torch.equal(idx32, table_of(lv_r.neighbor_idx, key[0]).idx32), (i, key)


def test_gate_mark_and_bounded_wait():
    """crfconv_gate_mark / crfconv_gate_wait (csrc/rows.hip): a disabled gate never waits; an enabled one holds its stream until an
    unconsumed mark exists -- one mark opens it once -- and for at most max_wait_us when none comes (counted as a timeout); three
    timeouts in a row switch it off.  The waiting stream is one that runs BESIDE the marking stream (CollatePipeline.runs_beside_current,
    the probe the pipeline itself uses): HIP maps streams onto a few hardware queues, and on a shared queue the mark stays behind the
    wait -- which is what the self-switch-off is for."""
    import time
    from crfconv_amd import _lib
    from crfconv_amd.graph import ptr, stream_ptr
    from crfconv_amd.data import CollatePipeline
    gate = torch.zeros(4, dtype=torch.int64, device=DEV)
    for _ in range(8):
        side = torch.cuda.Stream(priority=max(torch.cuda.Stream.priority_range()))
        if CollatePipeline.runs_beside_current(side):
            break
    else:
        pytest.skip('no stream that runs beside the current one')

    def wait(us):
        with torch.cuda.stream(side):
            _lib.call('crfconv_gate_wait', ptr(gate), us, stream_ptr())
    wait(50000)                                                    # disabled: returns at once
    torch.cuda.synchronize()
    assert gate.tolist() == [0, 0, 0, 0]
    gate[2] = 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wait(20000)                                                    # no mark: 20 ms, then ahead
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert 0.015 < dt < 0.2 and gate.tolist() == [0, 0, 1 | (1 << 8), 1], (dt, gate.tolist())
    wait(100000)                                                   # waits on the side stream ...
    time.sleep(0.005)
    assert not side.query()
    _lib.call('crfconv_gate_mark', ptr(gate), stream_ptr())        # ... until the caller's stream marks
    torch.cuda.synchronize()
    assert gate.tolist() == [1, 1, 1, 1]                           # opened: the run of timeouts is forgotten
    for k in range(3):                                             # the mark is consumed: three waits in a row time out ...
        wait(2000)
    torch.cuda.synchronize()
    assert gate.tolist() == [1, 1, 0, 4]                           # ... and the gate has switched itself off
    t0 = time.perf_counter()
    wait(100000)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.05 and gate.tolist() == [1, 1, 0, 4]


def test_multiscale_compute_fps_branch():
    """sample_method='fps' (datasets/semantic3d_dataset.py:520-523): per-cloud farthest-point subsets, first pick point 0,
    each later pick the farthest from the picks before it (numpy restatement); sub_pos / sub_idx gathered per cloud and
    up_idx = nearest sampled point."""
    import crfconv_amd
    B, N = 2, 600
    pos = np.stack([S.make_cloud(40 + b, N) for b in range(B)])
    data = crfconv_amd.multiscale_compute(t(pos), sample_method='fps', sort='none', num_scales=2, ratio=(4, 3))
    with pytest.raises(NotImplementedError):
        crfconv_amd.multiscale_compute(t(pos), sample_method='grid', num_scales=1)

    def np_fps(p, m):
        d = ((p - p[0]) ** 2).sum(1)
        picks = [0]
        for _ in range(m - 1):
            j = int(np.argmax(d))
            picks.append(j)
            d = np.minimum(d, ((p - p[j]) ** 2).sum(1))
        return np.array(picks)
    lvl0, lvl1 = data.multiscale
    assert lvl1.pos.shape == (B, N // 4, 3) and lvl0.sub_idx.shape == (B, N // 4, 16)
    for b in range(B):
        want = np_fps(pos[b].astype(np.float64), N // 4)
        got_pos = lvl1.pos[b].cpu().numpy()
        assert np.array_equal(got_pos, pos[b][want])
        assert np.array_equal(lvl0.sub_idx[b].cpu().numpy(), lvl0.neighbor_idx[b].cpu().numpy()[want])
        d = ((pos[b][:, None, :].astype(np.float64) - got_pos[None].astype(np.float64)) ** 2).sum(-1)
        assert np.array_equal(lvl0.up_idx[b, :, 0].cpu().numpy(), d.argmin(1))


def test_static_batch_load_refreshes_tables_reverse_csr_and_moments():
    """MultiScaleData.load_: a second batch copied into the first batch's tensors must give exactly what that batch
    gives when collated on its own -- logits, loss and every gradient -- although the neighbour tables, reverse CSRs and
    rel-pos moments of the first batch already exist (they are refreshed into the same buffers, which is what lets a
    captured hipGraph of the step be replayed on fresh batches).  Also the in-place edit of `pos` alone (jitter) must
    be noticed by the memoised BatchNorm-1 moments (version counters), not silently ignored."""
    import crfconv_amd
    from crfconv_amd import models, ops
    B, N = 2, 4096

    def collate(seed):
        pos = np.stack([S.make_cloud(seed + b, N, box=(2, 2, 1)) for b in range(B)])
        feats = np.concatenate([pos, S.uniform(seed, 'rgb', (B, N, 3), 0, 1)], -1)
        labels = S.integers(seed, 'y', (B, N), 0, 14)
        g = torch.Generator().manual_seed(seed)
        return crfconv_amd.multiscale_compute(t(pos), x=t(feats), y=t(labels), generator=g)

    net = models.PointConvBig(6, 13, True, 3)
    net.load_state_dict(S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 4))
    net = net.to(DEV).train()
    net.classifier[1] = nn.Identity()

    def run(data):
        for p in net.parameters():
            p.grad = None
        for mod in net.modules():                       # same BatchNorm running state for every run
            if isinstance(mod, nn.BatchNorm1d):
                mod.reset_running_stats()
        logits = net(data)
        loss = ops.training_loss(logits, data.y, None, ignore_index=-1)
        loss.backward()
        return logits.detach().clone(), float(loss.detach()), {k: v.clone() for k, v in grads(net).items()}

    a = collate(100)
    run(a)                                              # tables, reverse CSRs, moments of batch A now exist
    ref_logits, ref_loss, ref_grads = run(collate(200))
    a.load_(collate(200))
    got_logits, got_loss, got_grads = run(a)
    assert torch.equal(got_logits, ref_logits) and got_loss == ref_loss
    for k in ref_grads:      # (dW2 of a PointConv sums wavefront partials with LDS float atomics: last-bit order effects)
        assert_close(got_grads[k], ref_grads[k], 1e-6, k)
    # positions edited in place: the next forward must see them (it does not equal the stale result)
    a.multiscale[0].pos.mul_(1.5)
    moved_logits, _, _ = run(a)
    b = collate(200)
    b.multiscale[0].pos.mul_(1.5)
    fresh_logits, _, _ = run(b)
    assert torch.equal(moved_logits, fresh_logits) and not torch.equal(moved_logits, ref_logits)


# ------------------------------------------------------------------ full-size properties (config 2)
def test_meanfield_properties_full_size():
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    from crfconv_amd.utils import nearest_neighbors
    B, N, K, H, T = 4, 40960, 16, 8, 3
    g = torch.Generator().manual_seed(0)
    pos = (torch.rand(B, N, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])).to(DEV)
    nbr = nearest_neighbors.knn_batch_device(pos, pos, K)
    tab = NeighborTable(nbr, N)
    y = torch.randn(B * N, H, generator=g).to(DEV)
    z1 = torch.randn(B * N, H, generator=g).to(DEV)
    z2 = torch.randn(B * N, H, generator=g).to(DEV)
    c = (torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)).to(DEV)
    f = lambda z: ops.crf_meanfield(z, y, c, tab, T)
    # linear in z for fixed pairwise features
    lhs = f(0.3 * z1 - 1.7 * z2)
    rhs = 0.3 * f(z1) - 1.7 * f(z2)
    assert float((lhs - rhs).abs().max()) < 1e-4
    # constant field: A is row-stochastic, so z = 1 v^T stays rank one: x_T rows all equal
    v = torch.randn(H, generator=g).to(DEV)
    out = f(v.expand(B * N, H).contiguous())
    assert float((out - out[0]).abs().max()) < 1e-5
    # fixed point of the iteration for that field: x = (v + x C)(I + C)^-1  <=>  x = v
    assert float((out[0] - v).abs().max()) < 1e-5


def test_meanfield_single_step_inference_skips_s():
    """T = 1 without gradients takes the no-s-store form of the fused kernel: same x_1 as the training form."""
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    pos, nbr = knn_tables(2, 3000, 16, 77)
    tab = NeighborTable(t(nbr), 3000)
    g = torch.Generator().manual_seed(5)
    z, y = torch.randn(6000, 8, generator=g).to(DEV), torch.randn(6000, 8, generator=g).to(DEV)
    c = (torch.eye(8) + 0.1 * torch.randn(8, 8, generator=g)).to(DEV)
    with torch.no_grad():
        a = ops.crf_meanfield(z, y, c, tab, 1)
    b = ops.crf_meanfield(z.clone().requires_grad_(), y, c, tab, 1)
    assert torch.equal(a, b.detach())


@pytest.mark.parametrize('name,B,N,K,T,H', [('C2 S3DIS batch (headline)', 4, 40960, 16, 3, 8), ('C3 KITTI scan', 1, 122880, 16, 1, 8), ('C4 ScanNet cloud', 4, 81920, 16, 3, 8),
                                            ('C5 Semantic3D crops', 2, 65536, 32, 5, 8), ('C5 level 1', 2, 16384, 32, 5, 16)])
def test_meanfield_other_configs_full_size(name, B, N, K, T, H):
    """BASELINE.json configs 2-5 at full size: the mean-field forward AND backward against the CPU oracle on the whole
    batch (the oracle takes seconds at these sizes), through the int32 index path (clouds > 65536 points: C3, C4) and
    the uint16 path at its limit (65536-point crops, K = 32: C5)."""
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    from crfconv_amd.utils import nearest_neighbors
    g = torch.Generator().manual_seed(N + K)
    pos = (torch.rand(B, N, 3, generator=g) * torch.tensor([20.0, 20.0, 4.0])).to(DEV)
    nbr = nearest_neighbors.knn_batch_device(pos, pos, K)
    tab = NeighborTable(nbr, N)
    assert (tab.idx16 is not None) == (N <= 65536)
    z = torch.randn(B * N, H, generator=g).to(DEV).requires_grad_()
    y = (0.5 * torch.randn(B * N, H, generator=g)).to(DEV).requires_grad_()
    c = (torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)).to(DEV).requires_grad_()
    gout = torch.randn(B * N, H, generator=g).to(DEV)
    out = ops.crf_meanfield(z, y, c, tab, T)
    (out * gout).sum().backward()
    torch.set_num_threads(16)
    zc = z.detach().cpu().reshape(B, N, H).requires_grad_()
    yc = y.detach().cpu().reshape(B, N, H).requires_grad_()
    cc = c.detach().cpu().requires_grad_()
    ref = O.crf_meanfield(zc, yc, nbr.cpu()[:, :, 1:], cc, T)          # column 0 = the query itself, dropped by position
    (ref * gout.cpu().reshape(B, N, H)).sum().backward()
    assert_close(out, ref.reshape(B * N, H), OUT_TOL, name + ' forward')
    assert_close(z.grad, zc.grad.reshape(B * N, H), GRAD_TOL, name + ' dz')
    assert_close(y.grad, yc.grad.reshape(B * N, H), GRAD_TOL, name + ' dy')
    assert_close(c.grad, cc.grad, GRAD_TOL, name + ' dc')


def test_config5_shape_network_k32_t5_vs_oracle():
    """BASELINE.json config 5's operator shapes at a size the oracle trains in seconds: K = 32 at every level
    (kernel_size=[32]*5), five mean-field steps; eval logits, then loss and every parameter gradient in train mode."""
    import crfconv_amd
    from crfconv_amd import models
    B, N, ncls = 2, 8192, 8
    pos = np.stack([S.make_cloud(90 + b, N, box=(6.0, 6.0, 1.5)) for b in range(B)])
    feats = np.concatenate([pos, S.uniform(90, 'rgb', (B, N, 3), 0, 1)], -1).astype(np.float32)
    labels = S.integers(90, 'y', (B, N), 0, ncls + 1)
    choices, n = [], N
    for i, r in enumerate((4, 4, 4, 4, 2)):
        choices.append(torch.from_numpy(S.permutation(90, 'c%d' % i, n)[: n // r]))
        n //= r
    data = crfconv_amd.multiscale_compute(t(pos), x=t(feats), choices=choices, kernel_size=(32,) * 5)
    assert data.multiscale[0].neighbor_idx.shape == (B, N, 32) and data.multiscale[4].neighbor_idx.shape == (B, 32, 32)
    net = models.PointConvBig(6, ncls, use_crf=True, steps=5)
    sd = S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 11)
    net.load_state_dict(sd)
    net = net.to(DEV)
    ms = [{k: getattr(l, k).cpu() for k in ('pos', 'neighbor_idx', 'sub_idx', 'up_idx')} for l in data.multiscale]
    torch.set_num_threads(16)
    net.eval()
    with torch.no_grad():
        got = net(data)
        ref = O.pointconv_resnet({k: v.clone() for k, v in sd.items()}, torch.from_numpy(feats), ms, 5, False, True)
    assert_close(got, ref, OUT_TOL, 'K=32 T=5 eval logits')
    net.train()
    net.classifier[1] = nn.Identity()                      # dropout draws from different RNG streams: compare without
    logits = net(data)
    y = t(labels, torch.long).reshape(-1) - 1
    loss = torch.nn.functional.cross_entropy(logits, y, ignore_index=-1)
    loss.backward()
    prm = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    mask = torch.full((B, N, 128), 0.5)                    # oracle: h * mask * 2 = identity
    ref_t = O.pointconv_resnet(prm, torch.from_numpy(feats), ms, 5, True, True, dropout_mask=mask)
    ref_loss = O.training_loss(ref_t, torch.from_numpy(labels))
    ref_loss.backward()
    assert_close(logits, ref_t, 1e-4, 'K=32 T=5 train logits')      # measured 2.4e-6; was 5e-4
    assert_close(loss, ref_loss, 1e-4, 'loss')
    gr = grads(net)
    worst = max((relerr(gr[k], prm[k].grad), k) for k in gr)
    if os.environ.get('CRFCONV_TEST_REPORT'):
        print('[worst gradient] %s %.3e' % (worst[1], worst[0]), flush=True)
    assert worst[0] <= 2e-4, 'worst gradient %s: %.2e' % (worst[1], worst[0])      # measured 4.8e-6; was 3e-3


def _eval_net_vs_oracle(pos, feats, in_ch, ncls, steps, seed, name, g, ratio=(4, 4, 4, 4, 2), kernel_size=(16,) * 5):
    """Whole PointConvBig in eval mode on `pos` [B, N, 3] / `feats` [B, N, in_ch]: per-point logits within 1e-4 of the
    CPU oracle and the same arg-max labels ("mIoU parity": identical confusion matrix up to provably ambiguous rows)."""
    import crfconv_amd
    from crfconv_amd import models
    from crfconv_amd.utils import runningScore
    B, N = pos.shape[:2]
    data = crfconv_amd.multiscale_compute(pos.to(DEV), x=feats.to(DEV), generator=g, ratio=ratio, kernel_size=kernel_size)
    net = models.PointConvBig(in_ch, ncls, use_crf=True, steps=steps)
    sd = S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    with torch.no_grad():
        logits = net(data)
    assert logits.shape == (B * N, ncls) and torch.isfinite(logits).all()
    torch.set_num_threads(16)
    ms = [{k: getattr(l, k).cpu() for k in ('pos', 'neighbor_idx', 'sub_idx', 'up_idx') if getattr(l, k, None) is not None}
          for l in data.multiscale]
    with torch.no_grad():
        # the device collate emits each cloud in Morton order: the oracle gets the features in that same order
        ref = O.pointconv_resnet({k: v.clone() for k, v in sd.items()}, data.x.cpu(), ms, steps, False, True)
        # float64 run of the same oracle: with O(10^5) points a few dozen rows sit on ill-conditioned sums (the float32
        # ORACLE itself is 1e-4 .. 6e-4 off the float64 result there), so the 1e-4 bar is anchored on float64: every
        # row within max(1e-4, 4 x the float32 oracle's own error), and all but 0.1 % of the rows within 1e-4 outright
        ms64 = [{k: (v.double() if v.is_floating_point() else v) for k, v in l.items()} for l in ms]
        ref64 = O.pointconv_resnet({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()},
                                   data.x.cpu().double(), ms64, steps, False, True)
    _report_err(name + ' eval logits', logits, ref64)
    assert_close_anchored(logits, ref, ref64, OUT_TOL, name + ' logits')
    scale = max(1.0, float(ref64.abs().max()))
    # ABSOLUTE figures (north_star: "per-point logits within 1e-4"), recorded per BASELINE config (tests/golden/parity_report.json, read by
    # bench.py's `parity`).  Two weight sets: (a) the random-GAIN state dict above -- logits of magnitude 10^1 .. 10^5, where the float32
    # CPU oracle itself is 10^-2 .. 10^1 away from its float64 run: absolute error there is a statement about float32, and ours sits
    # beside the oracle's; (b) the constructor's own initialisation (logits of magnitude one, like a trained network's): there the
    # absolute bound of north_star applies as it stands and is asserted.
    def absolute(got, ref32, ref64_):
        abs_row = (got.detach().cpu().double() - ref64_).abs().max(1).values
        abs_row32 = (ref32.double() - ref64_).abs().max(1).values
        return {'max_abs': float(abs_row.max()), 'max_normalised': float(abs_row.max()) / max(1.0, float(ref64_.abs().max())),
                'max_abs_logit': float(ref64_.abs().max()), 'rows': int(abs_row.numel()), 'rows_beyond_1e-4_abs': int((abs_row > 1e-4).sum()),
                'f32_oracle_max_abs_vs_f64': float(abs_row32.max()), 'f32_oracle_rows_beyond_1e-4_abs': int((abs_row32 > 1e-4).sum())}
    rec = {'random_gain_weights': absolute(logits, ref, ref64)}
    torch.manual_seed(seed)
    net_d = models.PointConvBig(in_ch, ncls, use_crf=True, steps=steps)
    sd_d = {k: v.detach().clone() for k, v in net_d.state_dict().items()}
    net_d = net_d.to(DEV).eval()
    with torch.no_grad():
        logits_d = net_d(data)
        ref_d = O.pointconv_resnet({k: v.clone() for k, v in sd_d.items()}, data.x.cpu(), ms, steps, False, True)
        ref_d64 = O.pointconv_resnet({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd_d.items()},
                                     data.x.cpu().double(), ms64, steps, False, True)
    rec['default_init_weights'] = absolute(logits_d, ref_d, ref_d64)
    del net_d
    print('[parity] %s %s' % (name, rec), flush=True)
    if os.environ.get('CRFCONV_PARITY_RECORD'):
        import json
        path = os.environ['CRFCONV_PARITY_RECORD']
        try:
            allrec = json.load(open(path))
        except (OSError, ValueError):
            allrec = {}
        allrec[name] = rec
        json.dump(allrec, open(path, 'w'), indent=1, sort_keys=True)
    for which, r_ in rec.items():
        if r_['f32_oracle_max_abs_vs_f64'] <= 1e-4:        # wherever float32 itself can be within 1e-4, the kernels are -- in absolute terms
            assert r_['max_abs'] <= 1e-4, '%s (%s): max |logit - oracle| %.3e absolute although the float32 oracle is within %.3e of float64' % (
                name, which, r_['max_abs'], r_['f32_oracle_max_abs_vs_f64'])
    assert rec['default_init_weights']['max_abs'] <= 1e-4, rec['default_init_weights']      # logits of magnitude one: north_star's bound as it stands
    row_err = (logits.detach().cpu().double() - ref64).abs().max(1).values / scale
    assert float((row_err > OUT_TOL).double().mean()) <= 1e-3, '%s: %d rows beyond 1e-4' % (name, int((row_err > OUT_TOL).sum()))
    labels = torch.randint(0, ncls, (B * N,), generator=g)
    a, b = runningScore(ncls), runningScore(ncls)
    a.update_from_logits(labels.to(DEV), logits)
    b.update(labels.to(DEV), ref64.argmax(1).to(DEV))
    # arg-max may flip only where the two largest logits are closer than the error actually allowed above
    top2 = ref64.topk(2, dim=1).values
    margin = 2 * max(OUT_TOL, float(row_err.max())) * scale
    ambiguous = int(((top2[:, 0] - top2[:, 1]) < margin).sum())
    assert np.abs(a.confusion_matrix - b.confusion_matrix).sum() <= 2 * ambiguous
    return data, net, sd, ms


def test_config3_inference_matches_oracle():
    """BASELINE.json config 3: one KITTI-like scan of 122 880 points (ranges 2-50 m on 64 elevation rings), K = 16,
    one mean-field step, PointConvBig in eval mode."""
    g = torch.Generator().manual_seed(33)
    N, ncls = 122880, 19
    r = 2 + 48 * torch.rand(N, generator=g)
    az = 2 * np.pi * torch.rand(N, generator=g)
    el = torch.deg2rad(-25 + 28 * torch.randint(0, 64, (N,), generator=g).float() / 63)
    pos = torch.stack([r * torch.cos(el) * torch.cos(az), r * torch.cos(el) * torch.sin(az), r * torch.sin(el)], 1)
    pos = (pos + 0.01 * torch.randn(N, 3, generator=g)).float().unsqueeze(0)
    feats = torch.cat([pos, torch.rand(1, N, 1, generator=g)], -1)                      # xyz + remission: in_channels = 4
    _eval_net_vs_oracle(pos, feats, 4, ncls, 1, 9, 'config-3', g)


def _voxel_cloud(rng, n, dims, vox=0.04):
    flat = rng.choice(int(np.prod(dims)), size=n, replace=False)
    ijk = np.stack(np.unravel_index(flat, dims), -1).astype(np.float64)
    return ((ijk + 0.5) * vox + rng.uniform(-0.01, 0.01, (n, 3))).astype(np.float32)


def test_config4_whole_network_matches_oracle_full_size():
    """BASELINE.json config 4 at one GPU's share: 4 clouds x 81 920 points of a 6 x 6 x 3 m indoor box (one point per
    4 cm voxel), K = 16, three mean-field steps, 20 classes -- the WHOLE PointConvBig in eval mode against the CPU
    oracle (float32, anchored on float64), and the confusion matrix."""
    g = torch.Generator().manual_seed(4)
    rng = np.random.default_rng(4)
    B, N = 4, 81920
    pos = torch.from_numpy(np.stack([_voxel_cloud(rng, N, (150, 150, 75)) for _ in range(B)]))
    feats = torch.cat([pos, torch.rand(B, N, 3, generator=g)], -1)
    _eval_net_vs_oracle(pos, feats, 6, 20, 3, 14, 'config-4', g)


def test_config5_crop_whole_network_matches_oracle_full_size():
    """BASELINE.json config 5, ONE crop at full size: the 65 536 points of a 1 M-point-per-(60 x 60 x 15 m)-scene density
    nearest to a seed (semantic3d_dataset.py:433: a kNN ball), K = 32 at every level, FIVE mean-field steps, 8 classes,
    eval mode (B = 1) -- whole-network logits against the CPU oracle."""
    g = torch.Generator().manual_seed(5)
    N, ncls = 65536, 8
    # scene density: 1 048 576 points in 54 000 m^3 -> a ball of 65 536 points has radius ~9.3 m, clipped by the 15 m ceiling;
    # draw 3x the points in the bounding cylinder and keep the N nearest to the seed
    cand = torch.rand(6 * N, 3, generator=g) * torch.tensor([24.0, 24.0, 15.0]) - torch.tensor([12.0, 12.0, 7.5])
    near = cand.norm(dim=1).argsort()[:N]
    pos = cand[near].unsqueeze(0).contiguous()
    feats = torch.cat([pos, torch.rand(1, N, 3, generator=g)], -1)
    _eval_net_vs_oracle(pos, feats, 6, ncls, 5, 15, 'config-5 crop', g, kernel_size=(32,) * 5)


def test_config2_headline_inference_matches_oracle():
    """BASELINE.json config 2 -- the shape the bench line is quoted on: 4 clouds x 40 960 points, one point per 4 cm
    voxel of an 8 x 8 x 3 m box, K = 16, three mean-field steps, 13 classes.  Whole-network eval logits
    vs the CPU oracle at FULL size, and the confusion matrix."""
    g = torch.Generator().manual_seed(2)
    B, N = 4, 40960
    rng = np.random.default_rng(2)
    dims = np.array([200, 200, 75])
    pos = np.empty((B, N, 3), np.float32)
    for b in range(B):
        flat = rng.choice(int(dims.prod()), size=N, replace=False)
        ijk = np.stack(np.unravel_index(flat, dims), -1).astype(np.float64)
        pos[b] = ((ijk + 0.5) * 0.04 + rng.uniform(-0.01, 0.01, (N, 3))).astype(np.float32)
    pos = torch.from_numpy(pos)
    feats = torch.cat([pos, torch.rand(B, N, 3, generator=g)], -1)
    _eval_net_vs_oracle(pos, feats, 6, 13, 3, 12, 'config-2', g)


@pytest.mark.parametrize('cfg,B,N,ncls', [('config-2', 4, 40960, 13), ('config-4', 4, 81920, 20)])
def test_benchmarked_training_step_replayed_graph_matches_oracle(cfg, B, N, ncls):
    """The path bench.py TIMES, end to end, against the oracle at BASELINE config 2 (4 x 40 960 points, K = 16, T = 3) and at
    config 4's per-GPU share (4 x 81 920 points, 20 classes: the whole net in TRAIN mode, not only its components):
    `part_a` exactly as bench.py builds it -- zero grads, PointConvBig forward in train mode (BatchNorm-1 prefold, fork
    chain, classifier MLP -> Dropout -> Linear as ONE node with the counter-based mask), weighted cross entropy, backward
    under ``deferred_weight_grads(sink=bucket.view_of)``, ``bucket.pack()`` -- captured into a hipGraph and REPLAYED.  The
    oracle (trainval.py:99-106 on oracle/crf_oracle.py) gets the same dropout mask (ops.dropout_keep_mask: the mask is a
    function of (seed, num_batches_tracked, element)); logits, loss and the WHOLE flat gradient bucket are compared, the
    float32 oracle anchored on its own float64 run."""
    import bench
    import crfconv_amd
    from crfconv_amd import distributed as D
    from crfconv_amd import models, ops
    T = 3
    dev = torch.device('cuda', 0)
    gen = torch.Generator().manual_seed(77)
    data, _ = bench.make_batch(0, B, N, dev, gen, 'morton')
    torch.manual_seed(5)
    net = models.PointConvBig(6, ncls, use_crf=True, steps=T)
    sd = S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 31)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    assert type(net.classifier[1]) is nn.Dropout            # the fused MLP -> Dropout -> Linear node is what runs
    bucket = D.FlatGradAllReduce(net)
    cw = (0.5 + torch.rand(ncls, generator=gen)).to(dev)    # non-uniform class weights (configure.py:44-47 style)
    unit = torch.ones((), device=dev)
    keep = {}

    def part_a():
        bucket.zero()
        logits = net(data)
        keep['logits'] = logits
        loss = ops.training_loss(logits, data.y, cw, ignore_index=-1)
        with ops.deferred_weight_grads(sink=bucket.view_of):
            loss.backward(unit)
        bucket.pack()
        return loss.detach()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            part_a()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ga = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        static_loss = part_a()
    ga.replay()
    ga.replay()
    torch.cuda.synchronize()
    logits, loss, flat = keep['logits'].clone(), static_loss.clone(), bucket.flat.clone()
    bn = net.classifier[0].bn.batch_norm
    ctr = int(bn.num_batches_tracked)
    assert ctr >= 4                                         # 2 warm-up passes + 2 replays (the capture itself runs nothing): a device word
    mask = ops.dropout_keep_mask(ops.dropout_seed(32, 128), ctr, B * N * 128, 0.5).reshape(B, N, 128)
    kept = float(mask.mean())
    assert 0.499 < kept < 0.501
    # the mask the kernel drew: dropped elements of the classifier's hidden tensor are exact zeros
    # (checked through the oracle comparison below: a wrong mask moves the logits by O(1))
    ms = [{k: getattr(l, k).cpu() for k in ('pos', 'neighbor_idx', 'sub_idx', 'up_idx') if getattr(l, k, None) is not None}
          for l in data.multiscale]
    names = [k for k, p in net.named_parameters() if p.requires_grad]
    assert [id(p) for p in bucket.params] == [id(dict(net.named_parameters())[k]) for k in names]
    torch.set_num_threads(16)
    res = {}
    for tag, cast in (('f32', lambda v: v.clone()), ('f64', lambda v: v.double() if v.is_floating_point() else v.clone())):
        prm = {k: cast(v).requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
        msx = [{k: cast(v) for k, v in l.items()} for l in ms]
        ref_t = O.pointconv_resnet(prm, cast(data.x.cpu()), msx, T, True, True, dropout_mask=cast(torch.from_numpy(mask).float()))
        ref_loss = O.training_loss(ref_t, data.y.cpu(), cast(cw.cpu()))
        ref_loss.backward()
        res[tag] = (ref_t.detach(), ref_loss.detach(), torch.cat([prm[k].grad.reshape(-1) for k in names]).detach())
        del prm, msx, ref_t, ref_loss
    (t32, l32, g32), (t64, l64, g64) = res['f32'], res['f64']
    _report_err(cfg + ' replayed train logits', logits, t64)
    _report_err(cfg + ' replayed loss', loss, l64)
    _report_err(cfg + ' replayed flat gradient bucket', flat, g64)
    assert_close_anchored(logits, t32, t64, OUT_TOL, 'replayed train logits')       # the stated 1e-4 (measured: 2e-6 normalised, 7e-5 absolute)
    assert_close_anchored(loss, l32, l64, 1e-5, 'replayed loss')
    assert flat.numel() == g64.numel()
    # per parameter tensor, relative to that tensor's largest gradient entry
    o, worst = 0, (0.0, '')
    for k in names:
        n = dict(net.named_parameters())[k].numel()
        e = relerr(flat[o:o + n], g64[o:o + n])
        e32 = relerr(g32[o:o + n], g64[o:o + n])
        assert e <= max(5e-4, 4.0 * e32), '%s: gradient err %.2e (fp32 oracle %.2e)' % (k, e, e32)      # measured worst: 4e-5
        worst = max(worst, (e, k))
        o += n
    print('worst parameter gradient: %s %.2e' % (worst[1], worst[0]))


def test_config1_shape_eval_and_train_vs_oracle():
    """BASELINE.json config 1's exact shape: 2 x 2048 points on the unit sphere's interior, xyz + unit normals
    (in_channels 6), K = 16, ONE mean-field step, 50 part classes: eval logits + confusion, then train-mode logits (WITH the
    classifier's dropout: the oracle gets the mask the kernel drew), loss and every parameter gradient against the oracle."""
    g = torch.Generator().manual_seed(1)
    B, N, ncls = 2, 2048, 50
    pos = torch.rand(B, N, 3, generator=g) - 0.5
    pos = pos / pos.norm(dim=-1).max()
    nrm = torch.randn(B, N, 3, generator=g)
    nrm = nrm / nrm.norm(dim=-1, keepdim=True)
    feats = torch.cat([pos, nrm], -1)
    # 2048 points cannot carry the S3DIS ratios [4,4,4,4,2] with K = 16 (8 points would be left at level 4, fewer than
    # K -- the reference's kNN fails there too): the five levels halve instead, 2048 -> 128
    data, net, sd, ms = _eval_net_vs_oracle(pos, feats, 6, ncls, 1, 21, 'config-1', g, ratio=(2, 2, 2, 2, 2))
    labels = torch.randint(0, ncls + 1, (B, N), generator=g)
    net.train()
    assert type(net.classifier[1]) is nn.Dropout           # the classifier's dropout stays: its mask is a function of (seed, step counter,
    logits = net(data)                                     # element) at every size, so the oracle can be handed the same one
    loss = torch.nn.functional.cross_entropy(logits, labels.reshape(-1).to(DEV) - 1, ignore_index=-1)
    loss.backward()
    from crfconv_amd import ops
    ctr = int(net.classifier[0].bn.batch_norm.num_batches_tracked)
    keep = ops.dropout_keep_mask(ops.dropout_seed(32, 128), ctr, B * N * 128, 0.5).reshape(B, N, 128)
    assert 0.49 < float(keep.mean()) < 0.51
    prm = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    mask = torch.from_numpy(keep).float()
    ref_t = O.pointconv_resnet(prm, data.x.cpu(), ms, 1, True, True, dropout_mask=mask)
    ref_loss = O.training_loss(ref_t, labels)
    ref_loss.backward()
    assert_close(logits, ref_t, 1e-4, 'config-1 train logits')      # measured 2.6e-6; was 5e-4
    assert_close(loss, ref_loss, 1e-4, 'config-1 loss')
    gr = grads(net)
    worst = max((relerr(gr[k], prm[k].grad), k) for k in gr)
    if os.environ.get('CRFCONV_TEST_REPORT'):
        print('[worst gradient] %s %.3e' % (worst[1], worst[0]), flush=True)
    # 2048-point clouds: level 5 has 4 points per cloud, BatchNorm there is ill-conditioned for the float32 oracle too
    assert worst[0] <= 1e-3, 'worst gradient %s: %.2e' % (worst[1], worst[0])      # measured 1.8e-4; was 3e-3


@pytest.mark.parametrize('M,Ci,Co,bias', [(163840, 32, 128, False), (40960, 64, 16, False), (1000, 6, 8, False),
                                          (2560, 512, 256, False), (777, 128, 13, True), (163840, 8, 8, False)])
def test_linear_wgrad_mfma(M, Ci, Co, bias):
    """dW / db of the per-point Linear layers (MFMA kernel) against float64 torch."""
    from crfconv_amd import ops
    g = torch.Generator().manual_seed(M + Ci)
    x = torch.randn(M, Ci, generator=g).to(DEV).requires_grad_(True)
    W = torch.randn(Co, Ci, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(Co, generator=g).to(DEV).requires_grad_(True) if bias else None
    go = torch.randn(M, Co, generator=g).to(DEV)
    y = ops.linear(x, W, b)
    y.backward(go)
    assert_close(y, torch.nn.functional.linear(x, W, b), 1e-5, 'y')
    assert_close(W.grad, go.double().t() @ x.detach().double(), 2e-5, 'dW')     # relative to max |dW|
    assert_close(x.grad, go @ W.detach(), 1e-5, 'dX')
    if bias:
        assert_close(b.grad, go.double().sum(0), 2e-5, 'db')


@pytest.mark.parametrize('min_rows', [4096, 12288])
@pytest.mark.parametrize('M,Ci,Co,slope,need_dx', [(163840, 32, 8, 0.1, True), (163840, 6, 32, 0.1, False), (40960, 64, 16, 1.0, True),
                                                   (163840, 32, 128, 0.1, True), (10240, 128, 32, 0.1, True), (4100, 24, 64, 0.1, True),
                                                   (40963, 16, 64, 1.0, True), (10240, 32, 128, 1.0, True),
                                                   # wide layers of the 2 560-point level: narrower weight slabs (32 / 16 channels per workgroup)
                                                   (10240, 256, 128, 0.1, True), (10240, 128, 256, 1.0, True), (4100, 512, 64, 0.1, True), (4100, 64, 512, 0.1, True),
                                                   # coarse levels: the one-launch forward of csrc/mlp_small.hip (grid barrier)
                                                   (2560, 256, 64, 0.1, True), (2560, 64, 256, 1.0, True), (1280, 512, 128, 0.1, True),
                                                   (1280, 128, 512, 1.0, True), (4095, 512, 512, 0.1, True), (1000, 80, 192, 0.2, True),
                                                   (64, 16, 64, 0.1, True), (37, 144, 64, 1.0, False)])
def test_mlp_block_fused_backward(M, Ci, Co, slope, need_dx, min_rows, monkeypatch):
    """ops.mlp_block (Linear -> train-mode BatchNorm -> LeakyReLU as one node, csrc/linear.hip: mlp_bwd_p1 / finalize /
    dX with the BatchNorm-backward prologue) against float64 torch.  As in test_fused_batchnorm_lrelu the LeakyReLU
    branch of elements within rounding of 0 is taken from the kernel's own output."""
    from crfconv_amd import ops
    g = torch.Generator().manual_seed(M + Ci + Co)
    x = (torch.randn(M, Ci, generator=g) + 0.5).to(DEV).requires_grad_(need_dx)
    W = (torch.randn(Co, Ci, generator=g) / np.sqrt(Ci)).to(DEV).requires_grad_(True)
    go = torch.randn(M, Co, generator=g).to(DEV)
    bn = torch.nn.BatchNorm1d(Co)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Co, generator=g) + 0.5); bn.bias.copy_(torch.rand(Co, generator=g) * 0.6 - 0.3)
    bn = bn.to(DEV).train()
    monkeypatch.setattr(ops.state, 'mfma_min_rows', min_rows)      # 4096: the row-streaming forms from 4 100 rows on; 12 288: the shipped switch-over
    small = ops._mlp_small_ok(M, Ci, Co)
    if not ops.mlp_block_ok(x, W, None, bn, True):
        # between the one-launch kernel's co-residency limit and the switch-over no fused block applies (the layer runs as
        # Linear + BatchNorm nodes: covered by test_mlp_and_bn_semantics_match_oracle and the whole-network tests)
        assert min_rows == 12288 and 4096 <= M < 12288 and not small
        pytest.skip('no fused block for %d rows at a switch-over of %d' % (M, min_rows))
    assert small == (M < 4096) or (small and M < min_rows)
    out = ops.mlp_block(x, W, bn, slope)
    if small:
        assert '_MLPSmall' in out.grad_fn.next_functions[0][0].name()
        assert int(ops.gridsync_ws(DEV).abs().sum()) == 0          # no barrier gave up, and the words are back to zero
    out.backward(go)
    ref = torch.nn.BatchNorm1d(Co).to(DEV).double().train()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in torch.nn.BatchNorm1d(Co).state_dict().items()})
    with torch.no_grad():
        ref.weight.copy_(bn.weight.double()); ref.bias.copy_(bn.bias.double())
    xr = x.detach().double().requires_grad_(True)
    Wr = W.detach().double().requires_grad_(True)
    pre = ref(xr @ Wr.t())
    branch = torch.where(out.detach() > 0, 1.0, slope).double()
    yr = torch.where(out.detach() > 0, pre, slope * pre)
    pre.backward(go.double() * branch)
    assert_close(out, yr, 1e-5, 'out')
    assert_close(W.grad, Wr.grad, 2e-5, 'dW')
    assert_close(bn.weight.grad, ref.weight.grad, 2e-5, 'dgamma')
    assert_close(bn.bias.grad, ref.bias.grad, 2e-5, 'dbeta')
    if need_dx:
        assert_close(x.grad, xr.grad, 2e-5, 'dx')
    assert_close(bn.running_mean, ref.running_mean, 1e-6, 'running_mean')
    assert_close(bn.running_var, ref.running_var, 1e-5, 'running_var')
    assert int(bn.num_batches_tracked) == 1


def test_mlp_block_dropout_fused_mask_is_consistent():
    """ops.mlp_block_dropout: the classifier's MLP -> Dropout(0.5) as one node with a counter-based mask.  Every output is
    either 0 or twice the un-dropped activation, about half are kept, forward and backward use the SAME mask (gradients equal
    the reference's with that mask applied), the next step draws a different mask, and a captured graph draws a new one at
    every replay."""
    from crfconv_amd import ops
    M, Ci, Co, slope = 40960, 32, 128, 0.1
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, Ci, generator=g).to(DEV).requires_grad_(True)
    W = (torch.randn(Co, Ci, generator=g) / 6).to(DEV).requires_grad_(True)
    go = torch.randn(M, Co, generator=g).to(DEV)
    bn, bn_ref = (torch.nn.BatchNorm1d(Co).to(DEV).train() for _ in range(2))
    out = ops.mlp_block_dropout(x, W, bn, slope, 0.5)
    assert out is not None and int(bn.num_batches_tracked) == 1
    out.backward(go)
    got = [t.grad.clone() for t in (x, W, bn.weight, bn.bias)]
    for t in (x, W):
        t.grad = None
    act = ops.mlp_block(x, W, bn_ref, slope)                       # the same block without dropout
    keep = out.detach() != 0
    assert 0.49 < float(keep.float().mean()) < 0.51
    assert torch.equal(out.detach()[keep], (2.0 * act.detach())[keep])
    (act * keep * 2.0).backward(go)
    for name, a, b in zip(('dx', 'dW', 'dgamma', 'dbeta'), got, (x.grad, W.grad, bn_ref.weight.grad, bn_ref.bias.grad)):
        assert_close(a, b, 1e-6, name)
    out2 = ops.mlp_block_dropout(x.detach(), W.detach(), bn, slope, 0.5)          # next step: another mask
    assert 0.45 < float(((out2 != 0) != keep).float().mean()) < 0.55
    # under a captured graph the counter advances inside the graph: every replay has its own mask
    xs = x.detach().clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.mlp_block_dropout(xs, W.detach(), bn, slope, 0.5)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_out = ops.mlp_block_dropout(xs, W.detach(), bn, slope, 0.5)
    masks = []
    for _ in range(3):
        graph.replay()
        masks.append((static_out != 0).clone())
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])
    # below the row-streaming forms' switch-over the node runs all the same (the counter-based mask at every size)
    small = ops.mlp_block_dropout(x[:100].detach(), W.detach(), bn, slope, 0.5)
    assert small is not None and 0.35 < float((small != 0).float().mean()) < 0.65


def test_classifier_dropout_backward_folded_into_last_linear():
    """ops.mlp_dropout_linear: MLP -> Dropout -> Linear as one node whose backward masks the last Linear's input gradient while
    the MFMA kernel writes it (crfconv_linear_forward_dropout) instead of in a pass of its own.  Same mask (same seed, same
    counter), same arithmetic: logits and every gradient are bit-identical to ops.mlp_block_dropout followed by ops.linear;
    with deferred weight gradients the flushed (dW2, db2) match too."""
    from crfconv_amd import ops
    M, Ci, Co, C2, slope = 40960, 32, 128, 13, 0.1
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(M, Ci, generator=g).to(DEV)
    W0 = (torch.randn(Co, Ci, generator=g) / 6).to(DEV)
    W20 = (torch.randn(C2, Co, generator=g) / 11).to(DEV)
    b20 = torch.randn(C2, generator=g).to(DEV)
    go = torch.randn(M, C2, generator=g).to(DEV)
    res = []
    for fused in (True, False):
        x, W, W2, b2 = (v.clone().requires_grad_(True) for v in (x0, W0, W20, b20))
        bn = torch.nn.BatchNorm1d(Co).to(DEV).train()
        if fused:
            out = ops.mlp_dropout_linear(x, W, bn, slope, 0.5, W2, b2, recompute=False)
            assert out is not None and '_MLPDropoutLinear' in out.grad_fn.next_functions[0][0].name()
        else:
            out = ops.linear(ops.mlp_block_dropout(x, W, bn, slope, 0.5), W2, b2)
        assert int(bn.num_batches_tracked) == 1
        out.backward(go)
        res.append([out.detach().clone()] + [v.grad.clone() for v in (x, W, W2, b2, bn.weight, bn.bias)])
    for name, a, b in zip(('logits', 'dx', 'dW', 'dW2', 'db2', 'dgamma', 'dbeta'), *res):
        assert torch.equal(a, b), name
    # deferred weight gradients: (dW2, db2) of the fused node travel through the same batched reduction as _Linear's
    x, W = x0.clone().requires_grad_(True), W0.clone().requires_grad_(True)
    W2, b2 = torch.nn.Parameter(W20.clone()), torch.nn.Parameter(b20.clone())
    bn = torch.nn.BatchNorm1d(Co).to(DEV).train()
    with ops.deferred_weight_grads():
        ops.mlp_dropout_linear(x, W, bn, slope, 0.5, W2, b2, recompute=False).backward(go)
    assert_close(W2.grad, res[0][3], 1e-6, 'deferred dW2')
    assert_close(b2.grad, res[0][4], 1e-6, 'deferred db2')
    assert torch.equal(x.grad, res[0][1])
    assert ops.mlp_dropout_linear(x[:100], W, bn, slope, 0.5, W2, b2) is None      # below the MFMA row count: caller's path


@pytest.mark.parametrize('M,Ci,C2,bias', [(40960, 32, 13, True), (20001, 32, 13, True), (12288, 16, 8, False), (16400, 16, 16, True)])
def test_classifier_head_recompute(M, Ci, C2, bias):
    """ops._HeadRecompute (csrc/head.hip): the classifier MLP -> Dropout -> Linear without a stored [M, 4 C] tensor.  Logits, dX
    and the parameter gradients agree with the stored form (_MLPDropoutLinear: same products in the same order, same mask) to
    float32 rounding and with a float64 evaluation of the same function given the same mask
    (ops.dropout_keep_mask).  Ragged row counts, 16 / 32 inputs, with and without the last bias; deferred weight gradients
    with a flat-bucket sink land in the sink."""
    from crfconv_amd import ops
    Co, slope, p = 128, 0.1, 0.5
    g = torch.Generator().manual_seed(11 + Ci + C2)
    x0 = torch.randn(M, Ci, generator=g).to(DEV)
    W0 = (torch.randn(Co, Ci, generator=g) / 6).to(DEV)
    W20 = (torch.randn(C2, Co, generator=g) / 11).to(DEV)
    b20 = torch.randn(C2, generator=g).to(DEV) if bias else None
    gam0, bet0 = (torch.rand(Co, generator=g) + 0.5).to(DEV), (torch.randn(Co, generator=g) * 0.3).to(DEV)
    go = torch.randn(M, C2, generator=g).to(DEV)
    res = []
    for rec in (True, False):
        x, W, W2 = (v.clone().requires_grad_(True) for v in (x0, W0, W20))
        b2 = b20.clone().requires_grad_(True) if bias else None
        bn = torch.nn.BatchNorm1d(Co).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(gam0); bn.bias.copy_(bet0)
        out = ops.mlp_dropout_linear(x, W, bn, slope, p, W2, b2, recompute=rec)
        assert out is not None and ('_HeadRecompute' if rec else '_MLPDropoutLinear') in out.grad_fn.next_functions[0][0].name()
        out.backward(go)
        res.append(dict(logits=out.detach().clone(), dx=x.grad.clone(), dW=W.grad.clone(), dW2=W2.grad.clone(),
                        db2=b2.grad.clone() if bias else None, dgamma=bn.weight.grad.clone(), dbeta=bn.bias.grad.clone(),
                        rm=bn.running_mean.clone(), rv=bn.running_var.clone()))
    new, old = res
    # same products in the same order and the same mask; the statistic records are cut differently (other workgroup shapes), so
    # the BatchNorm coefficients may differ in the last bit
    assert_close(new['logits'], old['logits'], 2e-6, 'head logits vs stored form')
    assert_close(new['rm'], old['rm'], 1e-6, 'head running mean')
    assert_close(new['rv'], old['rv'], 1e-6, 'head running var')
    # float64 evaluation with the same mask
    keep = torch.from_numpy(ops.dropout_keep_mask(ops.dropout_seed(Ci, Co), 1, M * Co, p).reshape(M, Co)).to(DEV)
    xd, Wd, W2d = (v.double().requires_grad_(True) for v in (x0, W0, W20))
    gd, bd = gam0.double().requires_grad_(True), bet0.double().requires_grad_(True)
    b2d = b20.double().requires_grad_(True) if bias else None
    y = xd @ Wd.t()
    yn = (y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + 1e-5) * gd + bd
    h = torch.nn.functional.leaky_relu(yn, slope) * keep / (1 - p)
    ref = h @ W2d.t() + (b2d if bias else 0)
    ref.backward(go.double())
    ref64 = dict(logits=ref.detach(), dx=xd.grad, dW=Wd.grad, dW2=W2d.grad, db2=b2d.grad if bias else None, dgamma=gd.grad, dbeta=bd.grad)
    for k in ('logits', 'dx', 'dW', 'dW2', 'db2', 'dgamma', 'dbeta'):
        if new[k] is None:
            continue
        assert_close(new[k], ref64[k], 2e-5, 'head %s vs float64' % k)
        assert_close(new[k], old[k], 2e-5, 'head %s vs stored form' % k)
    # deferred weight gradients with a sink: every parameter gradient is written straight into the caller's storage
    x = x0.clone().requires_grad_(True)
    W, W2 = torch.nn.Parameter(W0.clone()), torch.nn.Parameter(W20.clone())
    b2 = torch.nn.Parameter(b20.clone()) if bias else None
    bn = torch.nn.BatchNorm1d(Co).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(gam0); bn.bias.copy_(bet0)
    params = [W, bn.weight, bn.bias, W2] + ([b2] if bias else [])
    store = {id(q): torch.full_like(q, float('nan')) for q in params}
    with ops.deferred_weight_grads(sink=lambda q: store.get(id(q))):
        ops.mlp_dropout_linear(x, W, bn, slope, p, W2, b2, recompute=True).backward(go)
    for q, k in zip(params, ('dW', 'dgamma', 'dbeta', 'dW2', 'db2')):
        assert q.grad.data_ptr() == store[id(q)].data_ptr(), k
        assert torch.equal(q.grad, new[k]), k
    assert torch.equal(x.grad, new['dx'])
    assert ops.mlp_dropout_linear(x0, W0[:64], torch.nn.BatchNorm1d(64).to(DEV).train(), slope, p, W20[:, :64], b20, recompute=True) is None


def test_pointconv_prefold_one_launch_equals_per_layer_folds():
    """ops.point_conv_prefold: BatchNorm-1 of all ten weight MLPs folded in ONE launch before the forward pass (the network does
    it in training mode) against the per-layer fold inside every PointConv: logits, every gradient and every BatchNorm buffer
    (running statistics, step counters) bit-identical."""
    import copy
    import crfconv_amd
    from crfconv_amd import models, ops
    B, N = 2, 4096
    pos = np.stack([S.make_cloud(120 + b, N, box=(2, 2, 1)) for b in range(B)])
    feats = np.concatenate([pos, S.uniform(120, 'rgb', (B, N, 3), 0, 1)], -1)
    data = crfconv_amd.multiscale_compute(t(pos), t(feats), generator=torch.Generator().manual_seed(6))
    labels = t(S.integers(120, 'y', (B, N), 0, 14))
    torch.manual_seed(4)
    net0 = models.PointConvBig(6, 13, use_crf=True, steps=2).to(DEV).train()
    res = []
    for pre in (True, False):
        net = copy.deepcopy(net0)
        ops.state.no_prefold = not pre
        try:
            torch.manual_seed(9)                                     # same dropout stream
            logits = net(data)
            ops.training_loss(logits, labels, None, ignore_index=-1).backward()
        finally:
            ops.state.no_prefold = False
        res.append((logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()},
                    {k: b.clone() for k, b in net.named_buffers()}))
    (l1, g1, b1), (l2, g2, b2) = res
    assert torch.equal(l1, l2)
    for k in g1:
        if 'point_conv' in k and 'weight_nn.1.lin' in k:
            assert_close(g1[k], g2[k], 1e-6, k)                      # dW2 of a narrow PointConv: LDS float atomics
        else:
            assert torch.equal(g1[k], g2[k]), k
    for k in b1:
        assert torch.equal(b1[k], b2[k]), k
    assert any('point_conv.weight_nn.0.bn.batch_norm.running_var' in k and not torch.equal(b1[k], torch.ones_like(b1[k])) for k in b1)


def test_end_of_pass_launch_carrying_the_crf_matrices_backward():
    """ops.deferred_weight_grads: the backward of the CRF layers' matrices rides in the MLP blocks' weight-gradient launch
    (crfconv_mlp_dw_jobs_hosting, the default) -- against the two launches (state.dw_hosts_mats off) and against the pass without
    deferral: every gradient of the network, bit for bit between the two deferred forms."""
    import copy
    import crfconv_amd
    from crfconv_amd import models, ops
    B, N = 2, 4096
    pos = np.stack([S.make_cloud(130 + b, N, box=(2, 2, 1)) for b in range(B)])
    feats = np.concatenate([pos, S.uniform(130, 'rgb', (B, N, 3), 0, 1)], -1)
    data = crfconv_amd.multiscale_compute(t(pos), t(feats), generator=torch.Generator().manual_seed(6))
    labels = t(S.integers(130, 'y', (B, N), 0, 14))
    torch.manual_seed(4)
    net0 = models.PointConvBig(6, 13, use_crf=True, steps=2).to(DEV).train()
    with torch.no_grad():
        for m in net0.modules():
            if hasattr(m, 'c') and isinstance(m.c, torch.nn.Parameter):
                m.c.add_(0.1 * torch.randn_like(m.c))               # off the identity: dc has no zeros to hide behind
    res = []
    for mode in ('riders', 'two launches', 'no deferral'):
        net = copy.deepcopy(net0)
        ops.state.dw_hosts_mats = mode == 'riders'
        try:
            torch.manual_seed(9)
            loss = ops.training_loss(net(data), labels, None, ignore_index=-1)
            if mode == 'no deferral':
                loss.backward()
            else:
                with ops.deferred_weight_grads():
                    loss.backward()
        finally:
            ops.state.dw_hosts_mats = True
        res.append({k: p.grad.clone() for k, p in net.named_parameters()})
    g1, g2, g3 = res
    n_c = 0
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
        if k.endswith('.c'):
            n_c += 1
            assert float(g1[k].abs().max()) > 0
            assert_close(g1[k], g3[k], 1e-5, k)
    assert n_c == 4


@pytest.mark.usefixtures('big_forms_from_4096')
def test_resnet_join_fused_equals_two_passes():
    """models.common.mlp_join: lin_out's BatchNorm + the residual add + LeakyReLU as ONE pass (crfconv_bn_apply_add, one
    autograd node) against bn_apply followed by add_lrelu: the same arithmetic operation for operation, so outputs and every
    gradient are bit-identical -- for a block with an identity skip, for one whose shortcut is an MLP, and for a strided
    block, whose shortcut MLP + neighbour max-pool is one node too (BatchNorm applied while the pool gathers)."""
    from crfconv_amd import models, ops
    from crfconv_amd.models.point_conv_big import ResNetBBlock
    import crfconv_amd
    B, N = 2, 4096
    pos = np.stack([S.make_cloud(300 + b, N, box=(2, 2, 1)) for b in range(B)])
    data = crfconv_amd.multiscale_compute(t(pos), generator=torch.Generator().manual_seed(2))
    lvl = data.multiscale[0]
    for cin, cout, strided in ((32, 32, False), (16, 32, False), (32, 64, True)):
        torch.manual_seed(cin)
        blk = ResNetBBlock(cin, cout).to(DEV).train()
        blk_pos = (lvl.pos, data.multiscale[1].pos) if strided else lvl.pos      # strided: shortcut MLP + max-pool as one node
        blk_idx = lvl.sub_idx if strided else lvl.neighbor_idx
        n_out = data.multiscale[1].pos.shape[1] if strided else N
        x0 = torch.randn(B, N, cin, generator=torch.Generator().manual_seed(9)).to(DEV)
        go = torch.randn(B, n_out, cout, generator=torch.Generator().manual_seed(10)).to(DEV)
        res = []
        for fused in (True, False):
            ops.state.no_join = not fused
            for p in blk.parameters():
                p.grad = None
            for m in blk.modules():                                  # same running statistics in both runs
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.reset_running_stats()
            x = x0.clone().requires_grad_(True)
            out = blk(x, blk_pos, blk_idx)
            if fused:
                names = set()
                stack = [out.grad_fn]
                while stack:
                    f = stack.pop()
                    if f is None or f in names:
                        continue
                    names.add(f)
                    stack.extend(g for g, _ in f.next_functions)
                # (the strided block's output level here has 2048 rows: there the join is folded into the one-launch MLP)
                assert any(('_MLPBlockJoin' if B * n_out >= ops.state.mfma_min_rows else '_MLPSmallJoin') in f.name() for f in names)
                assert any('_MLPBlockPool' in f.name() for f in names) == strided
            out.backward(go)
            res.append((out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}))
        ops.state.no_join = False
        (o1, gx1, gp1), (o2, gx2, gp2) = res
        assert torch.equal(o1, o2) and torch.equal(gx1, gx2)
        for k in gp1:
            if 'point_conv' in k and 'weight_nn.1.lin' in k:
                assert_close(gp1[k], gp2[k], 1e-6, k)                # dW2 of a narrow PointConv: LDS float atomics
            else:
                assert torch.equal(gp1[k], gp2[k]), k


@pytest.mark.usefixtures('big_forms_from_4096')
def test_resnet_fork_input_gradient_added_inside_the_block_backward():
    """models.common.mlp_fork: the input of a ResNet block feeds lin_in AND the shortcut; the shortcut's gradient comes back
    through an alias that lin_in's node returned and is added while lin_in's backward writes dX (crfconv_mlp_backward_add; the
    beta = 1 epilogue of the GEMM at the coarse levels) instead of in an accumulation pass of autograd's.  One float addition
    either way: outputs, the input gradient and every parameter gradient are bit-identical to the un-forked graph at the MFMA
    levels; at the coarse levels (vendor GEMM, different epilogue) within float rounding."""
    from crfconv_amd import ops
    from crfconv_amd.models.point_conv_big import ResNetBBlock
    import crfconv_amd
    B, N = 2, 4096
    pos = np.stack([S.make_cloud(310 + b, N, box=(2, 2, 1)) for b in range(B)])
    data = crfconv_amd.multiscale_compute(t(pos), generator=torch.Generator().manual_seed(3))
    cases = [(0, 32, 32, False), (0, 32, 64, True), (1, 64, 256, False), (0, 24, 32, False)]      # level 1: 2048 rows, one-launch MLP
    for lv, cin, cout, strided in cases:
        lvl, nxt = data.multiscale[lv], data.multiscale[lv + 1]
        n_in = lvl.pos.shape[1]
        torch.manual_seed(cin + cout)
        blk = ResNetBBlock(cin, cout).to(DEV).train()
        blk_pos = (lvl.pos, nxt.pos) if strided else lvl.pos
        blk_idx = lvl.sub_idx if strided else lvl.neighbor_idx
        n_out = nxt.pos.shape[1] if strided else n_in
        x0 = torch.randn(B, n_in, cin, generator=torch.Generator().manual_seed(19)).to(DEV)
        go = torch.randn(B, n_out, cout, generator=torch.Generator().manual_seed(20)).to(DEV)
        res = []
        for fork in (True, False):
            ops.state.no_fork = not fork
            try:
                for p in blk.parameters():
                    p.grad = None
                for m in blk.modules():
                    if isinstance(m, torch.nn.BatchNorm1d):
                        m.reset_running_stats()
                x = x0.clone().requires_grad_(True)
                xin = x * 1.0                                        # a non-leaf input, as inside the network
                out = blk(xin, blk_pos, blk_idx)
                seen, stack, adds = set(), [out.grad_fn], 0
                while stack:
                    f = stack.pop()
                    if f is None or f in seen:
                        continue
                    seen.add(f)
                    stack.extend(g for g, _ in f.next_functions)
                mul = [f for f in seen if f.name() == 'MulBackward0']
                assert len(mul) == 1
                users = sum(1 for f in seen for g, _ in f.next_functions if g is mul[0])
                assert users == (1 if fork else 2), (users, fork, lv, cin, cout, strided)   # forked: lin_in's node is the only consumer of the input
                out.backward(go)
            finally:
                ops.state.no_fork = False
            res.append((out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}))
        (o1, gx1, gp1), (o2, gx2, gp2) = res
        small = B * n_in < ops.state.mfma_min_rows
        grouped = small and cin != cout                     # round 5: lin_in and the shortcut of such a block run as ONE node (ops.mlp_group:
        if grouped:                                         # the tiled product with statistic records), the un-forked graph as two
            assert_close(o1, o2, 2e-6, 'out (coarse level, grouped vs one by one)')      # one-launch kernels: another summation order
        else:
            assert torch.equal(o1, o2)
        if small:
            assert_close(gx1, gx2, 1e-5 if grouped else 1e-6, 'dX (coarse level)')
        else:
            assert torch.equal(gx1, gx2), (lv, cin, cout, strided)
        for k in gp1:
            if grouped or ('point_conv' in k and 'weight_nn.1.lin' in k):
                assert_close(gp1[k], gp2[k], 1e-5 if grouped else 1e-6, k)
            else:
                assert torch.equal(gp1[k], gp2[k]), k


def test_mlp_small_one_launch_kernel_under_graph_replay():
    """The coarse-level one-launch forward synchronises its workgroups through device words that every launch must leave
    zero, and exchanges statistic records past L1: captured into a hipGraph and replayed back to back on CHANGING inputs
    (and between eager launches of the same kernel) every replay must reproduce the eager result bit for bit."""
    from crfconv_amd import ops
    M, Ci, Co = 2560, 256, 512
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(M, Ci, generator=g).to(DEV) for _ in range(4)]
    W = (torch.randn(Co, Ci, generator=g) / 16).to(DEV).requires_grad_(True)
    go = torch.randn(M, Co, generator=g).to(DEV)
    bn = torch.nn.BatchNorm1d(Co).to(DEV).train()
    x_static = xs[0].clone().requires_grad_(True)

    def step():
        W.grad = x_static.grad = bn.weight.grad = bn.bias.grad = None
        out = ops.mlp_block(x_static, W, bn, 0.1)
        out.backward(go)
        return out.detach(), x_static.grad, W.grad

    eager = []
    for x in xs:
        x_static.data.copy_(x)
        eager.append([t.clone() for t in step()])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step()
    for rep in range(3):
        for i, x in enumerate(xs):
            x_static.data.copy_(x)
            graph.replay()
            if i % 2:
                ops.mlp_block(xs[(i + 1) % 4], W.detach(), bn, 0.1)           # an eager launch of the same kernel in between
            for name, a, b in zip(('out', 'dx', 'dW'), outs, eager[i]):
                assert torch.equal(a, b), (rep, i, name, float((a - b).abs().max()))
    assert int(ops.gridsync_ws(DEV).abs().sum()) == 0


@pytest.mark.usefixtures('big_forms_from_4096')
@pytest.mark.parametrize('M,Ca,Cb,Co', [(163840, 32, 32, 32), (40960, 64, 64, 64), (10240, 64, 32, 128), (5000, 8, 24, 16)])
def test_mlp_block_cat_equals_block_on_concatenation(M, Ca, Cb, Co):
    """ops.mlp_block_cat([xa | xb]) (two operand pointers, separate input gradients) against ops.mlp_block on the
    materialised torch.cat: same kernels, same summation order -> outputs and every gradient bit-identical."""
    from crfconv_amd import ops
    g = torch.Generator().manual_seed(M + Ca)
    xa = torch.randn(M, Ca, generator=g).to(DEV).requires_grad_(True)
    xb = torch.randn(M, Cb, generator=g).to(DEV).requires_grad_(True)
    W = (torch.randn(Co, Ca + Cb, generator=g) / 8).to(DEV).requires_grad_(True)
    go = torch.randn(M, Co, generator=g).to(DEV)
    bns = [torch.nn.BatchNorm1d(Co).to(DEV).train() for _ in range(2)]
    assert ops.mlp_block_cat(xa, xb[:, :3], W, bns[0], True, 0.1) is None      # width not a multiple of 4: caller's own path
    out = ops.mlp_block_cat(xa, xb, W, bns[0], True, 0.1)
    assert out is not None
    out.backward(go)
    got = [t_.grad.clone() for t_ in (xa, xb, W)] + [bns[0].weight.grad.clone(), bns[0].bias.grad.clone()]
    for t_ in (xa, xb, W):
        t_.grad = None
    ref = ops.mlp_block(torch.cat([xa, xb], 1), W, bns[1], 0.1)
    ref.backward(go)
    want = [xa.grad, xb.grad, W.grad, bns[1].weight.grad, bns[1].bias.grad]
    assert torch.equal(out, ref)
    for name, a, b in zip(('dxa', 'dxb', 'dW', 'dgamma', 'dbeta'), got, want):
        assert torch.equal(a, b), name
    assert torch.equal(bns[0].running_var, bns[1].running_var)


@pytest.mark.parametrize('M,C,slope,training', [(163840, 32, 0.1, True), (40960, 8, 1.0, True), (1000, 512, 0.1, True),
                                                (777, 128, 0.1, False), (33, 1024, 1.0, True), (2560, 256, 0.1, True),
                                                (640, 512, 1.0, True), (4096, 64, 0.1, True), (4097, 64, 0.1, True),
                                                (2, 16, 0.1, True), (2560, 24, 0.1, True)])
def test_fused_batchnorm_lrelu(M, C, slope, training):
    """csrc/bn.hip against torch BatchNorm1d + LeakyReLU in float64.  The LeakyReLU branch of the handful of
    elements whose pre-activation is within fp32 rounding of 0 is taken from the kernel's own output sign, so the
    comparison is exact elsewhere (otherwise one flipped element moves a channel sum by O(|g|))."""
    from crfconv_amd import ops
    g = torch.Generator().manual_seed(C + M)
    x = (torch.randn(M, C, generator=g) * 2 + 3).to(DEV).requires_grad_(True)      # mean >> 0: shifted sums matter
    go = torch.randn(M, C, generator=g).to(DEV)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.rand(C, generator=g) * 0.6 - 0.3)
        bn.running_mean.copy_(torch.rand(C, generator=g) + 2.5); bn.running_var.copy_(torch.rand(C, generator=g) * 2 + 3)
    bn = bn.to(DEV)
    ref = torch.nn.BatchNorm1d(C).to(DEV).double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    bn.train(training); ref.train(training)
    y = ops.bn_act(x, bn, training, slope)
    y.backward(go)
    xr = x.detach().double().requires_grad_(True)
    pre_ref = ref(xr)
    branch = torch.where(y.detach() > 0, 1.0, slope).double()
    yr = torch.where(y.detach() > 0, pre_ref, slope * pre_ref)
    pre_ref.backward(go.double() * branch)
    assert_close(y, yr, 1e-5, 'y')
    assert_close(x.grad, xr.grad, 2e-5, 'dx')
    assert_close(bn.weight.grad, ref.weight.grad, 2e-5, 'dgamma')
    assert_close(bn.bias.grad, ref.bias.grad, 2e-5, 'dbeta')
    assert_close(bn.running_mean, ref.running_mean, 1e-6, 'running_mean')
    assert_close(bn.running_var, ref.running_var, 1e-6, 'running_var')
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.parametrize('M,Ci,Co,bias', [(163840, 32, 128, False), (40960, 64, 16, False), (5000, 6, 32, False),
                                          (4099, 128, 13, True), (10240, 128, 64, False), (8192, 8, 8, True)])
@pytest.mark.usefixtures('big_forms_from_4096')
def test_linear_forward_mfma_and_fused_stats(M, Ci, Co, bias):
    """linear.hip forward / dX kernels vs float64 torch, and BatchNorm fed from the GEMM epilogue records."""
    from crfconv_amd import ops
    g = torch.Generator().manual_seed(M + Ci + Co)
    x = (torch.randn(M, Ci, generator=g) + 0.5).to(DEV).requires_grad_(True)
    W = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(Co, generator=g).to(DEV).requires_grad_(True) if bias else None
    go = torch.randn(M, Co, generator=g).to(DEV)
    y, rec = ops.linear(x, W, b, want_stats=True)
    assert rec is not None                                   # these shapes take the MFMA path
    y.backward(go)
    yr = torch.nn.functional.linear(x.detach().double(), W.detach().double(), None if b is None else b.detach().double())
    assert_close(y, yr, 2e-6, 'y')
    assert_close(x.grad, go.double() @ W.detach().double(), 2e-6, 'dX')
    assert_close(W.grad, go.double().t() @ x.detach().double(), 2e-5, 'dW')
    if Co % 4 == 0:
        bn = torch.nn.BatchNorm1d(Co).to(DEV).train()
        ref = torch.nn.BatchNorm1d(Co).to(DEV).double().train()
        out = ops.bn_act(y.detach(), bn, True, 1.0, records=rec)
        assert_close(out, ref(yr), 1e-5, 'BN from records')
        assert_close(bn.running_var, ref.running_var, 1e-6, 'running_var from records')


@pytest.mark.parametrize('M,N,K,nk', [(640, 128, 512, 0), (2560, 512, 256, 0), (2560, 256, 512, 0), (640, 64, 64, 0), (10240, 128, 128, 0),
                                      (163840, 32, 32, 0), (2560, 32, 256, 1), (10240, 128, 256, 1), (1000, 36, 20, 0), (77, 12, 8, 1),
                                      (1, 4, 4, 0), (65, 68, 100, 1), (33, 132, 36, 0)])
def test_gemm_kernel_vs_float64(M, N, K, nk):
    """crfconv_gemm (csrc/gemm.hip: the coarse-level Linear forward and every dX / g_h1 product that used to be a vendor
    GEMM) against float64 torch, with and without the bias / addend epilogue, ragged tile edges included; bitwise
    reproducible; every tile shape (CRFCONV_GEMM_TILE is read once, so the shapes are forced through the row count)."""
    from crfconv_amd import _lib
    from crfconv_amd.ops import ptr, stream_ptr, _gemm
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = (torch.randn(N, K, generator=g) if nk else torch.randn(K, N, generator=g)).to(DEV)
    bias, add = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    Bd = B.double().t() if nk else B.double()
    ref = A.double() @ Bd
    scale = float(ref.abs().max())
    C = _gemm(A, B, nk=bool(nk))
    assert float((C.double() - ref).abs().max()) <= 2e-6 * scale
    C2 = _gemm(A, B, bias, add, nk=bool(nk))
    assert float((C2.double() - (ref + bias.double() + add.double())).abs().max()) <= 2e-6 * (scale + 8.0)
    assert torch.equal(C, _gemm(A, B, nk=bool(nk)))
    # addend aliasing the output (the in-place accumulate form)
    acc = add.clone()
    _lib.call('crfconv_gemm', ptr(A), ptr(B), None, ptr(acc), M, N, K, nk, ptr(acc), stream_ptr())
    torch.cuda.synchronize()
    assert float((acc.double() - (ref + add.double())).abs().max()) <= 2e-6 * (scale + 8.0)


@pytest.mark.parametrize('M,N,K,nk', [(10, 6, 8, 0), (10, 8, 6, 0), (2000, 13, 128, 1), (2000, 128, 13, 0), (333, 7, 5, 1), (65, 1, 1, 0),
                                      (100, 3, 130, 0)])
def test_gemm_odd_widths(M, N, K, nk):
    """Widths that are not multiples of 4 (the 13-class logits below 4096 rows, their dX): the element-wise form of the same
    kernel, bias / addend included; and the Linear layer built on it."""
    from crfconv_amd import _lib, ops
    assert _lib.load().crfconv_gemm_supported(M, N, K) == 1
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = (torch.randn(N, K, generator=g) if nk else torch.randn(K, N, generator=g)).to(DEV)
    bias, add = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    ref = A.double() @ (B.double().t() if nk else B.double())
    assert_close(ops._gemm(A, B, nk=bool(nk)), ref, 2e-6, 'product')
    assert_close(ops._gemm(A, B, bias, add, nk=bool(nk)), ref + bias.double() + add.double(), 2e-6, 'product + bias + addend')
    if nk:
        x, W = A.clone().requires_grad_(True), B.clone().requires_grad_(True)
        y = ops.linear(x, W, bias)
        y.backward(add)
        assert_close(y, ref + bias.double(), 2e-6, 'linear')
        assert_close(x.grad, add.double() @ B.double(), 2e-6, 'dX')


@pytest.mark.parametrize('M,N,K', [(10240, 128, 256), (1000, 64, 32), (37, 8, 4), (4100, 132, 36)])
def test_gemm_stats_records_feed_batchnorm(M, N, K):
    """crfconv_gemm_stats: the product equals crfconv_gemm's bit for bit, and the statistic records of its epilogue (one per 16-row
    group, ragged last group included) give the BatchNorm coefficients / running statistics of the float64 reference."""
    from crfconv_amd import _lib, ops
    from crfconv_amd.ops import ptr, stream_ptr
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, K, generator=g) + 0.3).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    y = torch.empty(M, N, device=DEV)
    nrec = _lib.load().crfconv_gemm_stat_records(M)
    rec = torch.full((nrec, N, 4), float('nan'), device=DEV)
    _lib.call('crfconv_gemm_stats', ptr(A), ptr(W), M, N, K, ptr(y), ptr(rec), stream_ptr())
    assert torch.equal(y, ops._gemm(A, W, nk=True))
    assert int(rec[:, :, 1].sum(0).min()) == M and int(rec[:, :, 1].sum(0).max()) == M and not bool(torch.isnan(rec).any())
    bn = nn.BatchNorm1d(N).to(DEV).train()
    ref = nn.BatchNorm1d(N).to(DEV).double().train()
    coef = torch.empty(4 * N, device=DEV)
    _lib.call('crfconv_bn_coef_from_nrecords', ptr(rec), nrec, M, N, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
              0.1, 1e-5, ptr(coef), stream_ptr())
    yr = ref(y.double())
    assert_close(coef[:N] * y + coef[N:2 * N], yr, 1e-5, 'BatchNorm from the product\'s records')
    assert_close(bn.running_mean, ref.running_mean, 1e-6, 'running_mean')
    assert_close(bn.running_var, ref.running_var, 1e-6, 'running_var')


@pytest.mark.parametrize('M,Ci,Co,slope,addend', [(2560, 256, 64, 0.1, False), (2560, 64, 256, 1.0, True), (640, 512, 128, 0.1, True),
                                                   (2561, 128, 512, 0.2, False), (100, 8, 4, 0.1, True), (3000, 36, 132, 1.0, False)])
def test_mlp_small_backward_two_launch_form(M, Ci, Co, slope, addend):
    """crfconv_mlp_small_backward (tile sums + the dX product with gY formed in its operand load) against float64 torch and against
    the form it replaces (crfconv_bn_backward + crfconv_gemm): gY, dX (+ addend), dgamma, dbeta."""
    from crfconv_amd import ops, _lib
    from crfconv_amd.ops import ptr, stream_ptr
    g = torch.Generator().manual_seed(M + Ci + Co)
    y = (torch.randn(M, Co, generator=g) * 1.5 + 0.3).to(DEV)
    gA = torch.randn(M, Co, generator=g).to(DEV)
    W = (torch.randn(Co, Ci, generator=g) / Co ** 0.5).to(DEV)
    add = torch.randn(M, Ci, generator=g).to(DEV) if addend else None
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).to(DEV), torch.randn(Co, generator=g).to(DEV)
    yd = y.double()
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    a = gamma.double() * rstd
    coef = torch.cat([a, beta.double() - a * mean, mean, rstd]).float().contiguous()
    # float64 reference of the BatchNorm + LeakyReLU backward, with the kernel's own LeakyReLU branch (float32 coefficients)
    pre32 = torch.addcmul(coef[Co:2 * Co], coef[:Co], y)
    g1 = gA.double() * torch.where(pre32 > 0, 1.0, slope).double()
    yh = (yd - coef[2 * Co:3 * Co].double()) * coef[3 * Co:].double()
    dbeta_r, dgamma_r = g1.sum(0), (g1 * yh).sum(0)
    gY_r = coef[:Co].double() * (g1 - dbeta_r / M - yh * dgamma_r / M)
    dX_r = gY_r @ W.double() + (0 if add is None else add.double())
    dgamma, dbeta = torch.empty(Co, device=DEV), torch.empty(Co, device=DEV)
    gY, dX = ops._small_bwd(gA, y, coef, W, add, slope, dgamma, dbeta, True)
    assert_close(gY, gY_r, 2e-5, 'gY')
    assert_close(dX, dX_r, 2e-5, 'dX')
    assert_close(dgamma, dgamma_r, 2e-5, 'dgamma')
    assert_close(dbeta, dbeta_r, 2e-5, 'dbeta')
    # the form it replaces
    gY0 = torch.empty_like(y)
    dg0, db0 = torch.empty(Co, device=DEV), torch.empty(Co, device=DEV)
    nb = _lib.load().crfconv_bn_workspace(M, Co)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _lib.call('crfconv_bn_backward', ptr(gA), ptr(y), ptr(coef), M, Co, 1, float(slope), ptr(gY0), ptr(dg0), ptr(db0), ptr(ws), nb, stream_ptr())
    assert_close(gY, gY0, 1e-5, 'gY vs bn_backward')
    assert_close(dX, ops._gemm(gY0, W, addend=add), 1e-5, 'dX vs bn_backward + gemm')
    assert torch.equal(ops._small_bwd(gA, y, coef, W, add, slope, dgamma, dbeta, True)[1], dX)       # reproducible
    # the one-launch form (the default: product workgroups wait inside the launch for the tile sums) against the two launches: every bit
    dg2, db2 = torch.empty(Co, device=DEV), torch.empty(Co, device=DEV)
    assert ops.state.small_bwd_one_launch
    ops.state.small_bwd_one_launch = False
    try:
        gY2, dX2 = ops._small_bwd(gA, y, coef, W, add, slope, dg2, db2, True)
    finally:
        ops.state.small_bwd_one_launch = True
    assert torch.equal(gY2, gY) and torch.equal(dX2, dX) and torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)
    assert int(ops.gridsync_ws(DEV).abs().sum()) == 0         # wait words left zero, no failure code


def test_mlp_small_backward_one_launch_four_jobs_replayed():
    """crfconv_mlp_small_backward_jobs_one_launch with four jobs of different shapes (1 .. 8 column slabs, narrow and wide product tiles),
    launched twenty times back to back and replayed from a hipGraph: bit-identical to crfconv_mlp_small_backward_jobs every time, the
    wait words zero afterwards."""
    import ctypes
    from crfconv_amd import ops, _lib
    from crfconv_amd.ops import ptr, stream_ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(77)
    shapes = [(10240, 128, 256), (2560, 256, 512), (640, 512, 64), (333, 36, 132)]

    def problem(M, Ci, Co):
        y = (torch.randn(M, Co, generator=g) * 1.5 + 0.3).to(DEV)
        gA = torch.randn(M, Co, generator=g).to(DEV)
        W = (torch.randn(Co, Ci, generator=g) / Co ** 0.5).to(DEV)
        add = torch.randn(M, Ci, generator=g).to(DEV)
        mean, var = y.double().mean(0), y.double().var(0, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        coef = torch.cat([rstd, -rstd * mean, mean, rstd]).float().contiguous()
        return y, gA, W, add, coef
    probs = [problem(*s_) for s_ in shapes]

    def outputs():
        return [(torch.full((M, Co), float('nan'), device=DEV), torch.full((M, Ci), float('nan'), device=DEV), torch.empty(Co, device=DEV),
                 torch.empty(Co, device=DEV), torch.empty(lib.crfconv_mlp_small_backward_workspace(M, Co), dtype=torch.uint8, device=DEV))
                for M, Ci, Co in shapes]

    def jobs_of(outs):
        jobs = (_lib.MlpBwdJob * 4)()
        for i, ((M, Ci, Co), (y, gA, W, add, coef), (gY, dX, dg, db, ws)) in enumerate(zip(shapes, probs, outs)):
            jobs[i] = _lib.MlpBwdJob(gA.data_ptr(), y.data_ptr(), coef.data_ptr(), W.data_ptr(), add.data_ptr(), M, Ci, Co, 1, 0.1,
                                     gY.data_ptr(), dX.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel())
        return jobs
    ref = outputs()
    _lib.call('crfconv_mlp_small_backward_jobs', ctypes.cast(jobs_of(ref), ctypes.c_void_p), 4, ptr(ops._ticket(DEV)), stream_ptr())
    ws = ops.gridsync_ws(DEV)
    for trial in range(20):
        got = outputs()
        _lib.call('crfconv_mlp_small_backward_jobs_one_launch', ctypes.cast(jobs_of(got), ctypes.c_void_p), 4, ptr(ops._ticket(DEV)), ptr(ws), stream_ptr())
        for r, o in zip(ref, got):
            for a, b in zip(r[:4], o[:4]):
                assert torch.equal(a, b), trial
    assert int(ws.abs().sum()) == 0
    got = outputs()
    jb = jobs_of(got)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _lib.call('crfconv_mlp_small_backward_jobs_one_launch', ctypes.cast(jb, ctypes.c_void_p), 4, ptr(ops._ticket(DEV)), ptr(ws), stream_ptr())
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        wsc = ops.gridsync_ws(DEV)
        _lib.call('crfconv_mlp_small_backward_jobs_one_launch', ctypes.cast(jb, ctypes.c_void_p), 4, ptr(ops._ticket(DEV)), ptr(wsc), stream_ptr())
    for trial in range(5):
        for o in got:
            o[0].fill_(float('nan'))
            o[1].fill_(float('nan'))
        graph.replay()
        for r, o in zip(ref, got):
            for a, b in zip(r[:4], o[:4]):
                assert torch.equal(a, b), ('replay', trial)
    assert int(wsc.abs().sum()) == 0 and int(ws.abs().sum()) == 0


@pytest.mark.parametrize('shared', [False, True])
def test_mlp_group_equals_the_separate_coarse_blocks(shared):
    """ops.mlp_group (round 5): two / three independent coarse-level Linear + BatchNorm + LeakyReLU blocks as ONE node -- one product
    launch with statistic records and one apply launch forward, two launches backward -- against the same blocks run one by one
    (ops.mlp_block: _MLPSmall): outputs, running statistics, input gradients (incl. the fork alias and, shared, the summed gradient of
    the common input), parameter gradients immediate and deferred."""
    from crfconv_amd import ops
    g = torch.Generator().manual_seed(17 + shared)
    shapes = [(2560, 128, 64, 0.1, True), (2560, 128, 256, 1.0, False)] if shared else [(640, 512, 64, 0.1, False), (2560, 256, 64, 0.1, True), (10240, 32, 32, 1.0, False)]
    x0 = torch.randn(shapes[0][0], shapes[0][1], generator=g).to(DEV)

    def build():
        blocks = []
        for i, (m, ci, co, slope, fork) in enumerate(shapes):
            x = (x0 if (shared or i == 0) else torch.randn(m, ci, generator=torch.Generator().manual_seed(100 + i)).to(DEV)).clone().requires_grad_(True)
            lin = torch.nn.Linear(ci, co, bias=False).to(DEV)
            bn = torch.nn.BatchNorm1d(co).to(DEV).train()
            with torch.no_grad():
                lin.weight.copy_(torch.randn(co, ci, generator=torch.Generator().manual_seed(200 + i)) / ci ** 0.5)
                bn.weight.copy_(torch.rand(co, generator=torch.Generator().manual_seed(300 + i)) + 0.5)
                bn.bias.copy_(torch.randn(co, generator=torch.Generator().manual_seed(400 + i)))
            blocks.append([x, lin, bn, slope, fork])
        if shared:
            blocks[1][0] = blocks[0][0]
        return blocks

    def run(blocks, group, defer):
        gouts = [torch.randn(b[0].shape[0], b[1].out_features, generator=torch.Generator().manual_seed(500 + i)).to(DEV) for i, b in enumerate(blocks)]
        galias = torch.randn(blocks[0][0].shape if shared else blocks[1][0].shape, generator=torch.Generator().manual_seed(600)).to(DEV)
        if group:
            res = ops.mlp_group([(x, lin.weight, bn, slope, fork) for x, lin, bn, slope, fork in blocks], shared=shared)
            assert res is not None
        else:
            res = [ops.mlp_block(x, lin.weight, bn, slope, fork=fork) if fork else ops.mlp_block(x, lin.weight, bn, slope) for x, lin, bn, slope, fork in blocks]
        loss, outs = 0.0, []
        for r, b, go in zip(res, blocks, gouts):
            o, alias = r if b[4] else (r, None)
            outs.append(o)
            loss = loss + (o * go).sum()
            if alias is not None:
                loss = loss + (alias * galias).sum()
        with ops.deferred_weight_grads(enabled=defer):
            loss.backward()
        xs = [b[0] for b in blocks[:1]] if shared else [b[0] for b in blocks]
        return outs, [x.grad for x in xs], [(b[1].weight.grad, b[2].weight.grad, b[2].bias.grad, b[2].running_mean.clone(), b[2].running_var.clone()) for b in blocks]

    ref = run(build(), False, False)
    for defer in (False, True):
        got = run(build(), True, defer)
        for i, (a, b) in enumerate(zip(got[0], ref[0])):
            assert_close(a, b, 2e-6, 'out %d' % i)
        for i, (a, b) in enumerate(zip(got[1], ref[1])):
            assert_close(a, b, 1e-5, 'dx %d' % i)
        for i, (pa, pb) in enumerate(zip(got[2], ref[2])):
            for name, a, b in zip(('dW', 'dgamma', 'dbeta', 'running_mean', 'running_var'), pa, pb):
                assert_close(a, b, 1e-5, '%s %d' % (name, i))
    # the node is used where it should be: strided coarse ResNet blocks and the coarse CRF layers of the network
    from crfconv_amd import models
    assert ops.mlp_group([(torch.zeros(163840, 32, device=DEV), torch.zeros(16, 32, device=DEV), torch.nn.BatchNorm1d(16).to(DEV), 0.1, False)] * 2) is None


@pytest.mark.parametrize('m,ca,cb', [(2560, 256, 256), (10240, 128, 128), (7, 4, 12), (1, 8, 4)])
def test_cat2_and_its_backward(m, ca, cb):
    """ops.cat2 (rows.hip) == torch.cat, forward and backward (contiguous gradients), 3-D leading shape kept."""
    from crfconv_amd import ops
    g = torch.Generator().manual_seed(m + ca)
    xa = torch.randn(1, m, ca, generator=g).to(DEV).requires_grad_(True)
    xb = torch.randn(1, m, cb, generator=g).to(DEV).requires_grad_(True)
    go = torch.randn(1, m, ca + cb, generator=g).to(DEV)
    out = ops.cat2(xa, xb)
    assert out.shape == (1, m, ca + cb) and torch.equal(out, torch.cat([xa, xb], -1))
    out.backward(go)
    assert xa.grad.is_contiguous() and xb.grad.is_contiguous()
    assert torch.equal(xa.grad, go[..., :ca]) and torch.equal(xb.grad, go[..., ca:])
    # widths that are not multiples of 4: the framework's cat
    assert ops.cat2(xa[..., :3], xb).shape == (1, m, 3 + cb)


def test_bucket_pack_and_counters_use_library_launches():
    """FlatGradAllReduce.pack() (crfconv_copy_jobs, more pairs than one launch takes) and advance_counters (crfconv_add_i64)."""
    from crfconv_amd import distributed as D, ops
    net = torch.nn.ModuleList([torch.nn.Linear(3 + i % 5, 2 + i % 3) for i in range(60)]).to(DEV)     # 120 parameters > 96 jobs
    bucket = D.FlatGradAllReduce(net)
    grads = []
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(i + 1)) if i % 7 else None
        grads.append(None if p.grad is None else p.grad.clone())
    bucket.pack()
    for p, v, g0 in zip(bucket.params, bucket.views, grads):
        assert p.grad.data_ptr() == v.data_ptr()
        assert torch.equal(v, torch.zeros_like(v) if g0 is None else g0)
    bns = torch.nn.Sequential(*[torch.nn.BatchNorm1d(4) for _ in range(5)]).to(DEV).train()
    for _ in range(3):
        with ops.advance_counters(bns):
            pass
    assert [int(b.num_batches_tracked) for b in bns] == [3] * 5


@pytest.mark.parametrize('weighted', [True, False])
def test_training_loss_matches_oracle(weighted):
    """trainval.py:101-104: weighted CE over 1-based labels with 0 = unlabeled (ignored)."""
    from crfconv_amd import ops
    m, C = 5000, 13
    logits = t(S.uniform(7, 'logits', (m, C)) * 6).requires_grad_()
    labels = t(S.integers(7, 'labels', (m,), 0, C + 1))                       # 0 -> ignored
    w = t(np.abs(S.uniform(7, 'w', (C,))) + 0.2) if weighted else None
    loss = ops.training_loss(logits, labels, w, ignore_index=-1)
    (loss * 1.7).backward()
    lc = logits.detach().cpu().requires_grad_()
    ref = O.training_loss(lc, labels.cpu(), None if w is None else w.cpu(), -1)
    (ref * 1.7).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-6 * max(1.0, abs(float(ref.detach())))
    assert_close(logits.grad, lc.grad, 1e-5, 'dlogits')
    assert float(logits.grad[labels == 0].abs().max()) == 0.0                 # ignored rows get no gradient


def test_cross_entropy_edge_cases():
    from crfconv_amd import ops
    from crfconv_amd._lib import CrfConvError
    z = t(S.uniform(8, 'z', (300, 5)))
    y = t(S.integers(8, 'y', (300,), 0, 5))
    ref = torch.nn.functional.cross_entropy(z.cpu(), y.cpu())
    assert abs(float(ops.cross_entropy(z, y)) - float(ref)) < 1e-6
    # every row ignored -> nan, like the framework
    assert torch.isnan(ops.cross_entropy(z, torch.full_like(y, -100)))
    # one row
    assert abs(float(ops.cross_entropy(z[:1], y[:1])) - float(torch.nn.functional.cross_entropy(z[:1].cpu(), y[:1].cpu()))) < 1e-6
    with pytest.raises(CrfConvError):
        ops.cross_entropy(z, y[:10])
    with pytest.raises(CrfConvError):
        ops.cross_entropy(z.cpu(), y.cpu())


@pytest.mark.parametrize('H', [1, 5, 8, 13, 16, 32, 47, 64])
def test_spd_inverse(H):
    """(I + c^T c)^-1 on one workgroup (continuous_crf_conv_big.py:72) against float64 LAPACK, plus its gradient."""
    from crfconv_amd import ops
    c = S.uniform(H, 'c', (H, H)).astype(np.float64) * 0.7 + np.eye(H)
    M = np.eye(H) + c.T @ c
    Mt = t(M.astype(np.float32)).requires_grad_()
    Q = ops._SpdInverse.apply(Mt)
    ref = np.linalg.inv(M.astype(np.float32).astype(np.float64))
    assert_close(Q, torch.from_numpy(ref).float(), 2e-6, 'Q')
    g = S.uniform(H, 'g', (H, H))
    (Q * t(g)).sum().backward()
    assert_close(Mt.grad, torch.from_numpy(-(ref.T @ g.astype(np.float64) @ ref.T)).float(), 1e-5, 'dM')
    with pytest.raises(Exception):
        ops._SpdInverse.apply(torch.eye(65, device=DEV))


@pytest.mark.parametrize('H', [1, 8, 13, 32, 64])
def test_crf_matrices(H):
    """Q = (I + c^T c)^-1, P = c^T c Q and dc, against float64 autograd."""
    from crfconv_amd import ops
    c0 = S.uniform(H + 100, 'c', (H, H)) * 0.4 + np.eye(H, dtype=np.float32)
    c = t(c0).requires_grad_()
    Q, P = ops._CrfMatrices.apply(c)
    gq, gp = S.uniform(H, 'gq', (H, H)), S.uniform(H, 'gp', (H, H))
    ((Q * t(gq)).sum() + (P * t(gp)).sum()).backward()
    cd = torch.from_numpy(c0).double().requires_grad_()
    C = cd.t() @ cd
    Qr = torch.linalg.inv(torch.eye(H, dtype=torch.float64) + C)
    Pr = C @ Qr
    ((Qr * torch.from_numpy(gq).double()).sum() + (Pr * torch.from_numpy(gp).double()).sum()).backward()
    assert_close(Q, Qr.float(), 2e-6, 'Q')
    assert_close(P, Pr.float(), 2e-6, 'P')
    assert_close(c.grad, cd.grad.float(), 1e-5, 'dc')
    # only one of the outputs used
    c2 = t(c0).requires_grad_()
    ops._CrfMatrices.apply(c2)[1].sum().backward()
    cd.grad = None
    (cd.t() @ cd @ torch.linalg.inv(torch.eye(H, dtype=torch.float64) + cd.t() @ cd)).sum().backward()
    assert_close(c2.grad, cd.grad.float(), 1e-5, 'dc from P only')


@pytest.mark.parametrize('cout,act', [(20, True), (20, False), (13, True)])
def test_mlp_and_bn_semantics_match_oracle(cout, act):
    """MLP = Linear -> FastBatchNorm1d -> activation (models/common.py:26-40), train and eval, fused and unfused
    (cout % 4 != 0) routes, against the oracle's mlp."""
    from crfconv_amd.models import MLP
    m = MLP(12, cout, activation=torch.nn.LeakyReLU(0.1) if act else None)
    sd = S.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 2)
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = S.uniform(2, 'x', (3, 50, 12))
    prm = {k: v.clone() for k, v in sd.items()}
    m.train()
    a = m(t(x))
    b = O.mlp(prm, '', torch.from_numpy(x), True, 0.1 if act else None)
    assert_close(a, b, 1e-5, 'train')
    assert_close(m.bn.batch_norm.running_var, prm['bn.batch_norm.running_var'], 1e-6, 'running_var')
    assert int(m.bn.batch_norm.num_batches_tracked) == 1
    m.eval()
    assert_close(m(t(x)), O.mlp(prm, '', torch.from_numpy(x), False, 0.1 if act else None), 1e-5, 'eval')


def test_deferred_weight_grads_equal_immediate():
    """ops.deferred_weight_grads(): all Linear dW / db reductions of a backward pass in one batched launch must give
    the same parameter gradients as the immediate path (same partials, same summation order), also when .grad
    accumulates over two passes."""
    import crfconv_amd
    from crfconv_amd import models, ops
    B, N = 2, 4096
    pos = np.stack([S.make_cloud(90 + b, N, box=(2, 2, 1)) for b in range(B)])
    feats = np.concatenate([pos, S.uniform(90, 'rgb', (B, N, 3), 0, 1)], -1)
    data = crfconv_amd.multiscale_compute(t(pos), t(feats), generator=torch.Generator().manual_seed(5))
    labels = t(S.integers(90, 'y', (B, N), 0, 14))
    net = models.PointConvBig(6, 13, use_crf=True, steps=2).to(DEV).train()
    net.classifier[1] = FixedDropout(torch.ones(B, N, 128, device=DEV) * 0.5)

    def run(defer, passes):
        for p in net.parameters():
            p.grad = None
        for _ in range(passes):
            loss = ops.training_loss(net(data), labels, None, ignore_index=-1)
            with ops.deferred_weight_grads(defer):
                loss.backward()
        return {k: p.grad.clone() for k, p in net.named_parameters()}

    ref2 = run(False, 1)
    noise = {k: float((v - w).abs().max()) for (k, v), w in zip(run(False, 1).items(), ref2.values())}
    for passes in (1, 2):
        a, b = run(False, passes), run(True, passes)
        assert set(a) == set(b)
        for k in a:
            # same partials and the same summation order; what is left is the run-to-run noise of the vendor GEMMs
            # upstream (measured above between two immediate runs)
            tol = 4 * noise[k] * passes + 1e-6 * float(a[k].abs().max())
            assert float((a[k] - b[k]).abs().max()) <= tol, (k, float((a[k] - b[k]).abs().max()), tol)
    # outside the context nothing is pending and autograd.grad still sees the weights
    w = net.conv1_1.lin_in.lin.weight
    (gw,) = torch.autograd.grad(ops.training_loss(net(data), labels, None, ignore_index=-1), [w])
    assert torch.isfinite(gw).all() and float(gw.abs().max()) > 0


@pytest.mark.usefixtures('big_forms_from_4096')
def test_deferred_weight_grads_into_the_flat_bucket():
    """ops.deferred_weight_grads(sink=bucket.view_of): the batched dW / db reduction writes straight into the flat
    gradient bucket.  Same values as without the sink, every Linear weight's .grad IS its bucket slice afterwards, pack()
    leaves the bucket equal to the gradients, and a second (accumulating) pass still adds up."""
    import crfconv_amd
    from crfconv_amd import distributed as D, models, ops
    B, N = 2, 4096
    pos = np.stack([S.make_cloud(95 + b, N, box=(2, 2, 1)) for b in range(B)])
    feats = np.concatenate([pos, S.uniform(95, 'rgb', (B, N, 3), 0, 1)], -1)
    data = crfconv_amd.multiscale_compute(t(pos), t(feats), generator=torch.Generator().manual_seed(5))
    labels = t(S.integers(95, 'y', (B, N), 0, 14))
    net = models.PointConvBig(6, 13, use_crf=True, steps=2).to(DEV).train()
    net.classifier[1] = FixedDropout(torch.ones(B, N, 128, device=DEV) * 0.5)
    bucket = D.FlatGradAllReduce(net)

    def run(sink, passes=1):
        bucket.zero()
        for _ in range(passes):
            loss = ops.training_loss(net(data), labels, None, ignore_index=-1)
            with ops.deferred_weight_grads(sink=sink):
                loss.backward()
        return {k: p.grad for k, p in net.named_parameters()}

    ref = {k: v.clone() for k, v in run(None).items()}
    got = run(bucket.view_of)
    direct = 0
    for (k, p), v in zip(net.named_parameters(), bucket.views):
        assert float((got[k] - ref[k]).abs().max()) <= 1e-6 * float(ref[k].abs().max()) + 1e-9, k
        direct += int(p.grad.data_ptr() == v.data_ptr())
    assert direct >= 60              # every Linear weight, and (dW, dgamma, dbeta) of the fused MLP blocks, went straight into the bucket
    fused_bn = [k for k, p in net.named_parameters() if k.startswith('conv1_2.lin_in.bn')]      # a fused block of level 0
    assert fused_bn and all(dict(net.named_parameters())[k].grad.data_ptr() == bucket.view_of(dict(net.named_parameters())[k]).data_ptr()
                            for k in fused_bn)
    bucket.pack()
    for (k, p), v in zip(net.named_parameters(), bucket.views):
        assert p.grad.data_ptr() == v.data_ptr() and torch.equal(v, got[k])
    two = run(bucket.view_of, passes=2)                   # second pass: .grad exists -> accumulate through a temporary
    for k in ref:
        assert float((two[k] - 2 * ref[k]).abs().max()) <= 1e-5 * float(ref[k].abs().max()) + 1e-9, k


@pytest.mark.parametrize('cfg', [dict(momentum=0.95, weight_decay=1e-4), dict(momentum=0.0, weight_decay=0.0),
                                 dict(momentum=0.9, weight_decay=1e-3, nesterov=True), dict(momentum=0.9, dampening=0.1)])
def test_flat_sgd_matches_torch_sgd(cfg):
    """optim.FlatSGD (one launch over flat parameters) against torch.optim.SGD, 4 steps, same gradients."""
    from crfconv_amd import optim
    from crfconv_amd.distributed import FlatGradAllReduce
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.BatchNorm1d(33), torch.nn.Linear(33, 5)).to(DEV)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.BatchNorm1d(33), torch.nn.Linear(33, 5)).to(DEV)
    ref.load_state_dict(net.state_dict())
    bucket = FlatGradAllReduce(net)
    opt = optim.FlatSGD(bucket, lr=0.05, **cfg)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.05, **cfg)
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))   # re-homing kept values
    for step in range(4):
        grads = [t(S.uniform(step, 'g%d' % i, tuple(p.shape))) for i, p in enumerate(ref.parameters())]
        for p, v, g in zip(ref.parameters(), bucket.views, grads):
            p.grad = g.clone()
            v.copy_(g)
        opt.step()
        ropt.step()
        for (k, a), b in zip(net.named_parameters(), ref.parameters()):
            assert_close(a, b, 1e-6, 'step %d %s' % (step, k))
    assert net[0].weight.data_ptr() == opt.flat.data_ptr()


def test_shared_weight_gradients_sum_inside_the_sink():
    """A parameter used TWICE in one backward under ``deferred_weight_grads(sink=...)``: the first use writes straight into the
    caller's slice, the second is added to it (both uses once aimed at the same slice and raced)."""
    from crfconv_amd import ops
    g = torch.Generator().manual_seed(3)
    M, Ci, Co = 8192, 16, 32
    W = nn.Parameter(torch.randn(Co, Ci, generator=g).to(DEV))
    b = nn.Parameter(torch.randn(Co, generator=g).to(DEV))
    x1 = torch.randn(M, Ci, generator=g).to(DEV)
    x2 = torch.randn(M, Ci, generator=g).to(DEV)
    slab = {id(W): torch.zeros(Co, Ci, device=DEV), id(b): torch.zeros(Co, device=DEV)}
    y = ops.linear(x1, W, b).sum() + 2.0 * ops.linear(x2, W, b).sum()
    with ops.deferred_weight_grads(sink=lambda p: slab.get(id(p))):
        y.backward()
    assert W.grad.data_ptr() == slab[id(W)].data_ptr() and b.grad.data_ptr() == slab[id(b)].data_ptr()
    ones = torch.ones(M, Co, device=DEV, dtype=torch.float64)
    assert_close(W.grad, ones.t() @ x1.double() + 2.0 * ones.t() @ x2.double(), 2e-5, 'shared dW')
    assert_close(b.grad, torch.full((Co,), 3.0 * M, dtype=torch.float64), 1e-6, 'shared db')


def test_crf_matrices_riding_in_the_first_pointconv_launch_equal_the_launch_of_their_own():
    """ops.crf_matrices_batched(cs, ride=True): the (Q, P) launch of the CRF layers is queued and CARRIED by the next PointConv
    statistics pass of a hosting width (d = 8: uvstats_hosting_kernel, the riders' workgroups first in the grid) -- Q, P, the
    PointConv output, every gradient must equal the two separate launches bit for bit.  A queued launch nobody carries (a d = 16
    PointConv) is issued by flush_riders() / by the first mean-field call."""
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    g = torch.Generator().manual_seed(21)
    n, K = 3000, 16
    idx = torch.randint(0, n, (1, n, K), generator=g)
    idx[0, :, 0] = torch.arange(n)
    table = NeighborTable(idx.to(DEV), n)
    pos = torch.rand(n, 3, generator=g).to(DEV)
    res = {}
    for d in (8, 16):
        for ride in (False, True):
            cs = [nn.Parameter((torch.eye(H) + 0.1 * torch.randn(H, H, generator=torch.Generator().manual_seed(H))).to(DEV)) for H in (64, 32, 16, 8)]
            gg = torch.Generator().manual_seed(d)
            x = torch.randn(n, d, generator=gg).to(DEV).requires_grad_(True)
            W1, W2 = nn.Parameter(torch.randn(d, 3, generator=gg).to(DEV)), nn.Parameter((torch.randn(d, d, generator=gg) / d ** 0.5).to(DEV))
            bn1, bn2 = nn.BatchNorm1d(d).to(DEV).train(), nn.BatchNorm1d(d).to(DEV).train()
            mats = ops.crf_matrices_batched(cs, ride=ride)
            assert (ops._RIDERS['mats'] is not None) == ride
            out = ops.point_conv(x, pos, pos, table, W1, bn1, W2, bn2, True)
            assert (ops._RIDERS['mats'] is not None) == (ride and d != 8)       # carried by the d = 8 launch only
            if d == 8:
                ops.flush_riders()                                               # nothing left: no launch
            else:
                z = torch.randn(n, 8, generator=gg).to(DEV)
                ops.crf_meanfield(z, z, cs[3], table, 1, matrices=mats[3])       # the first consumer flushes
                assert ops._RIDERS['mats'] is None
            loss = (out * out).sum() + sum((Q * Q).sum() + (P * Q).sum() for Q, P in mats)
            loss.backward()
            res[d, ride] = [out.detach()] + [t_.detach() for m in mats for t_ in m] + [x.grad, W1.grad, W2.grad] + [c.grad for c in cs]
        for a, b in zip(res[d, False], res[d, True]):
            assert float(a.abs().max()) > 0 and torch.equal(a, b)


@pytest.mark.parametrize('n', [20000, 40000, 40960])
def test_riders_in_a_statistics_pass_of_many_workgroups_leave_the_tickets_clean(n):
    """The hosting launch at sizes where the host's workgroups draw their tickets in groups (more than 64 workgroups) and their number
    is NOT a multiple of the group count (20 000 points = 157 workgroups, 40 000 = 313): the ticket group of a workgroup must come from
    its number among the host's workgroups, not from blockIdx.x behind the riders -- otherwise no workgroup is ever 'last', the
    BatchNorm-2 statistics are never written and the stream's ticket words stay dirty for every later launch.
    Output, running statistics and gradients must equal the launch without riders bit for bit; every ticket word must be zero."""
    from crfconv_amd import ops
    from crfconv_amd.ops._base import _ticket
    from crfconv_amd.graph import NeighborTable
    g = torch.Generator().manual_seed(n)
    K, d = 16, 8
    idx = torch.randint(0, n, (1, n, K), generator=g)
    idx[0, :, 0] = torch.arange(n)
    table = NeighborTable(idx.to(DEV), n)
    pos = torch.rand(n, 3, generator=g).to(DEV)
    res = {}
    for ride in (False, True):
        cs = [nn.Parameter((torch.eye(H) + 0.1 * torch.randn(H, H, generator=torch.Generator().manual_seed(H))).to(DEV)) for H in (64, 32, 16, 8)]
        gg = torch.Generator().manual_seed(d)
        x = torch.randn(n, d, generator=gg).to(DEV).requires_grad_(True)
        W1, W2 = nn.Parameter(torch.randn(d, 3, generator=gg).to(DEV)), nn.Parameter((torch.randn(d, d, generator=gg) / d ** 0.5).to(DEV))
        bn1, bn2 = nn.BatchNorm1d(d).to(DEV).train(), nn.BatchNorm1d(d).to(DEV).train()
        mats = ops.crf_matrices_batched(cs, ride=ride)
        out = ops.point_conv(x, pos, pos, table, W1, bn1, W2, bn2, True)
        assert ops._RIDERS['mats'] is None
        (out * out).sum().backward()
        torch.cuda.synchronize()
        assert int(_ticket(torch.device(DEV)).abs().sum()) == 0, 'ticket words left non-zero (ride=%s)' % ride
        res[ride] = [out.detach(), bn2.running_mean.clone(), bn2.running_var.clone(), x.grad, W1.grad, W2.grad] + [t_.detach() for m in mats for t_ in m]
    for a, b in zip(res[False], res[True]):
        assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0 and torch.equal(a, b)


def test_crf_parameter_gradients_deferred_to_the_end_of_the_pass_equal_the_immediate_ones():
    """Inside ops.deferred_weight_grads the CRF layers' dP / dQ (H >= 32: partial passes + sums in the batched launches) and the
    batched matrices backward run at the END of the backward pass and install dc as .grad; values must equal the immediate form's
    bit for bit (same kernels, same summation order), the activation gradients too."""
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    res = {}
    for mode in ('now', 'late'):
        cs = [nn.Parameter((torch.randn(H, H, generator=torch.Generator().manual_seed(H)) * 0.1).to(DEV)) for H in (16, 32, 64)]
        tabs, zs, ys = [], [], []
        for H, n in zip((16, 32, 64), (4096, 2560, 640)):
            gg = torch.Generator().manual_seed(n)
            idx = torch.randint(0, n, (1, n, 16), generator=gg)
            idx[0, :, 0] = torch.arange(n)
            tabs.append(NeighborTable(idx.to(DEV), n))
            zs.append(torch.randn(n, H, generator=gg).to(DEV).requires_grad_(True))
            ys.append(torch.randn(n, H, generator=gg).to(DEV).requires_grad_(True))

        def run():
            mats = ops.crf_matrices_batched(cs)
            loss = 0
            for c, mat, tb, z, y in zip(cs, mats, tabs, zs, ys):
                o = ops.crf_meanfield(z, y, c, tb, 3, k0=1, matrices=mat)
                loss = loss + (o * torch.linspace(0, 1, o.numel(), device=DEV).reshape(o.shape)).sum()
            return loss
        if mode == 'late':
            with ops.deferred_weight_grads():
                run().backward()
        else:
            run().backward()
        torch.cuda.synchronize()
        res[mode] = [c.grad.clone() for c in cs] + [z.grad.clone() for z in zs] + [y.grad.clone() for y in ys]
    for a, b in zip(res['now'], res['late']):
        assert float(a.abs().max()) > 0 and torch.equal(a, b)


@pytest.mark.parametrize('defer', [False, True])
def test_captured_step_of_the_unchanged_reference_loop_equals_the_eager_loop(defer):
    """crfconv_amd.train.CapturedStep around the reference's own step (trainval.py:99-106: zero_grad, model(data),
    F.cross_entropy(weight, ignore_index), backward, torch.optim.SGD.step) against the same loop run eagerly: three steps on
    three different batches from equal initial state must leave equal parameters, BatchNorm buffers and losses.  defer: the
    class's default -- the backward's weight-gradient launches batched at its end (ops.deferred_weight_grads inside the capture;
    another summation order of the same partial slabs) -- or the backward exactly as written (bit-equal to the eager loop)."""
    import crfconv_amd
    import torch.nn.functional as F
    from crfconv_amd import models
    from crfconv_amd.train import CapturedStep
    B, N = 2, 8192

    def batch(seed):
        pos = np.stack([S.make_cloud(seed + b, N, box=(2, 2, 1)) for b in range(B)])
        feats = np.concatenate([pos, S.uniform(seed, 'rgb', (B, N, 3), 0, 1)], -1)
        return crfconv_amd.multiscale_compute(t(pos), x=t(feats), y=t(S.integers(seed, 'y', (B, N), 0, 14)),
                                              generator=torch.Generator().manual_seed(seed))
    batches = [batch(900 + 10 * i) for i in range(3)]
    cw = torch.linspace(0.5, 1.5, 13, device=DEV)

    def loss_fn(out, d):
        return F.cross_entropy(out, d.y.reshape(-1) - 1, weight=cw, ignore_index=-1)
    torch.manual_seed(5)
    ref = models.PointConvBig(6, 13, True, 3).to(DEV).train()
    net = models.PointConvBig(6, 13, True, 3).to(DEV).train()
    net.load_state_dict(ref.state_dict())
    mk = lambda m: torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)      # noqa: E731
    ropt, opt = mk(ref), mk(net)
    ref_losses = []
    for d in batches:                                        # the reference loop, eagerly
        ropt.zero_grad()
        loss = loss_fn(ref(d), d)
        loss.backward()
        ropt.step()
        ref_losses.append(float(loss))
    static = batch(900)                                      # the resident batch the graph reads
    step = CapturedStep(net, opt, loss_fn, static, defer_weight_grads=defer)
    got_losses = [float(step(d)) for d in batches]
    torch.cuda.synchronize()
    for a, b in zip(got_losses, ref_losses):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (got_losses, ref_losses)
    for (k, a), b in zip(net.state_dict().items(), ref.state_dict().values()):
        assert_close(a.float(), b.float(), 2e-5, ('after 3 steps (batched weight gradients): ' if defer else 'after 3 steps: ') + k,
                     tighten=not defer)     # (batched weight gradients sum in another order: two trajectories, see assert_close)


def test_graphed_model_in_the_unchanged_reference_loop_equals_the_eager_loop():
    """crfconv_amd.train.GraphedModel: the reference's five lines VERBATIM (trainval.py:99-106) on a model wrapped once -- every
    model(data) one forward replay, loss.backward() one backward replay, F.cross_entropy and torch.optim.SGD the caller's own
    eager code -- against the same loop on the bare model: three batches from equal initial state leave equal losses, parameters
    and BatchNorm buffers.  Then: gradient ACCUMULATION over two backward passes without zero_grad (the wrapper's static gradient
    buffers must not alias what the caller holds), an eval-mode call and a no_grad call (both the wrapped model's eager path)."""
    import crfconv_amd
    import torch.nn.functional as F
    from crfconv_amd import models
    from crfconv_amd.train import GraphedModel
    B, N = 2, 8192

    def batch(seed):
        pos = np.stack([S.make_cloud(seed + b, N, box=(2, 2, 1)) for b in range(B)])
        feats = np.concatenate([pos, S.uniform(seed, 'rgb', (B, N, 3), 0, 1)], -1)
        return crfconv_amd.multiscale_compute(t(pos), x=t(feats), y=t(S.integers(seed, 'y', (B, N), 0, 14)),
                                              generator=torch.Generator().manual_seed(seed))
    batches = [batch(700 + 10 * i) for i in range(3)]
    cw = torch.linspace(0.5, 1.5, 13, device=DEV)
    torch.manual_seed(6)
    ref = models.PointConvBig(6, 13, True, 3).to(DEV).train()
    inner = models.PointConvBig(6, 13, True, 3).to(DEV).train()
    inner.load_state_dict(ref.state_dict())
    net = GraphedModel(inner)
    mk = lambda m: torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)      # noqa: E731
    losses = {}
    for name, model, optimizer in (('eager', ref, mk(ref)), ('graphed', net, mk(net))):
        losses[name] = []
        for data in batches:
            optimizer.zero_grad()
            y_pred = model(data)
            y = data.y.reshape(-1) - 1
            loss = F.cross_entropy(y_pred, y, weight=cw, ignore_index=-1)
            loss.backward()
            optimizer.step()
            losses[name].append(float(loss.detach()))
    assert net.fwd_graph is not None and net.bwd_graph is not None
    # the gradients autograd installed ARE the replay's static buffers (handed over as fresh tensor objects: no copy launch per parameter)
    assert all(q.grad is not None and q.grad.data_ptr() == sg.data_ptr() for q, sg in zip(net.params, net.static_grads) if sg is not None)
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys()) and net.C == ref.C       # transparent for checkpoints / attributes
    for a, b in zip(losses['graphed'], losses['eager']):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), losses
    for (k, a), b in zip(inner.state_dict().items(), ref.state_dict().values()):
        assert_close(a.float(), b.float(), 2e-5, 'graphed module, after 3 steps: ' + k, tighten=False)
    # accumulation: two backward passes on two batches, no zero_grad in between (from EQUAL state: what remains is summation order)
    inner.load_state_dict(ref.state_dict())
    seen = []
    hook = inner.conv1_1.lin_in.lin.weight.register_hook(lambda g: seen.append(g.clone()))      # a parameter hook sees each pass's gradient
    for model in (ref, net):
        for p in model.parameters():
            p.grad = None
        for data in batches[:2]:
            F.cross_entropy(model(data), data.y.reshape(-1) - 1, weight=cw, ignore_index=-1).backward()
    hook.remove()
    assert len(seen) == 2 and float(seen[0].abs().max()) > 0
    assert_close(seen[0] + seen[1], inner.conv1_1.lin_in.lin.weight.grad, 1e-6, 'graphed module, hook gradients sum to .grad')
    for (k, a), b in zip(inner.named_parameters(), ref.parameters()):
        assert_close(a.grad, b.grad, 2e-5, 'graphed module, accumulated gradient: ' + k)
    # a batch of another shape: the wrapped model runs eagerly, the captured graphs stay what they were
    other = crfconv_amd.multiscale_compute(t(np.stack([S.make_cloud(990 + b, N // 2, box=(2, 2, 1)) for b in range(B)])),
                                           x=t(S.uniform(991, 'f', (B, N // 2, 6), 0, 1)), y=t(S.integers(992, 'y', (B, N // 2), 0, 14)),
                                           generator=torch.Generator().manual_seed(9))
    inner.load_state_dict(ref.state_dict())
    outs = []
    for model in (net, ref):
        torch.manual_seed(123)          # (8 192 rows: the classifier runs the module's own nn.Dropout here -- Philox draws of the global generator)
        outs.append(model(other))
    a, b = outs
    assert a.shape == b.shape == (B * (N // 2), 13) and a.grad_fn is not None
    assert_close(a, b, 2e-5, 'graphed module, other shape (eager path)')
    # eval / no_grad: the wrapped model itself
    with torch.no_grad():
        a, b = net(batches[0]), ref(batches[0])
    assert_close(a, b, 2e-5, 'graphed module, no_grad call')
    net.eval(), ref.eval()
    a, b = net(batches[1]), ref(batches[1])
    assert_close(a, b, 2e-5, 'graphed module, eval call')


def test_bare_model_captures_itself_in_the_unchanged_reference_loop():
    """Round 6: with crfconv_amd.train's autograph on (the product's default; the suite runs with it off), the reference's five lines
    VERBATIM on the BARE model -- nothing wrapped -- are two hipGraph replays per step from the second iteration on: three batches from
    equal state leave the losses / parameters / buffers of the launch-by-launch loop.  Everything that does not fit runs eagerly and
    correctly: a forward whose predecessor has not been through backward() (its output must survive), eval / no_grad, another batch
    shape, a model with a forward hook; a deep copy starts without graphs and captures for itself; to() drops the graphs."""
    import copy
    import crfconv_amd
    import torch.nn.functional as F
    from crfconv_amd import models, train
    B, N = 2, 8192

    def batch(seed, n=N):
        pos = np.stack([S.make_cloud(seed + b, n, box=(2, 2, 1)) for b in range(B)])
        feats = np.concatenate([pos, S.uniform(seed, 'rgb', (B, n, 3), 0, 1)], -1)
        return crfconv_amd.multiscale_compute(t(pos), x=t(feats), y=t(S.integers(seed, 'y', (B, n), 0, 14)),
                                              generator=torch.Generator().manual_seed(seed))
    batches = [batch(700 + 10 * i) for i in range(3)]
    cw = torch.linspace(0.5, 1.5, 13, device=DEV)
    torch.manual_seed(6)
    ref = models.PointConvBig(6, 13, True, 3).to(DEV).train()
    net = models.PointConvBig(6, 13, True, 3).to(DEV).train()
    net.load_state_dict(ref.state_dict())
    mk = lambda m: torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)      # noqa: E731

    def loop(model, optimizer):
        out = []
        for data in batches:
            optimizer.zero_grad()
            y_pred = model(data)
            y = data.y.reshape(-1) - 1
            loss = F.cross_entropy(y_pred, y, weight=cw, ignore_index=-1)
            loss.backward()
            optimizer.step()
            out.append(float(loss.detach()))
        return out
    assert not train._AUTO['on'], 'the suite runs with autograph off (tests/conftest.py)'
    eager = loop(ref, mk(ref))
    assert '_autograph' not in ref.__dict__
    train.set_autograph(True)
    try:
        auto = loop(net, mk(net))
        runner = net.__dict__.get('_autograph')
        assert runner is not None and runner.fwd_graph is not None and runner.bwd_graph is not None
        assert list(net.state_dict().keys()) == list(ref.state_dict().keys())        # the runner is not part of the model's tree
        for a, b in zip(auto, eager):
            assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (auto, eager)
        for (k, a), b in zip(net.state_dict().items(), ref.state_dict().values()):
            assert_close(a.float(), b.float(), 2e-5, 'self-capturing model, after 3 steps: ' + k, tighten=False)
        # two forwards before one backward: the first output must survive the second call (which therefore runs eagerly)
        net.load_state_dict(ref.state_dict())
        for p in list(net.parameters()) + list(ref.parameters()):
            p.grad = None
        o1 = net(batches[0])
        keep = o1.detach().clone()
        o2 = net(batches[1])
        assert o2.grad_fn is not None and torch.equal(o1.detach(), keep), 'a replay overwrote an output that was still in use'
        (o1.sum() + o2.sum()).backward()
        with train.no_autograph():
            (ref(batches[0]).sum() + ref(batches[1]).sum()).backward()
        for (k, a), b in zip(net.named_parameters(), ref.parameters()):
            assert_close(a.grad, b.grad, 2e-5, 'self-capturing model, two forwards one backward: ' + k)
        # after that backward the replays are back
        before = runner.fwd_graph
        o3 = net(batches[2])
        assert type(o3.grad_fn).__name__.startswith('_GraphedPass') and runner.fwd_graph is before
        o3.sum().backward()
        # a step that skips its backward (say, a NaN loss): the next call runs eagerly while the orphaned output is alive, then replays return
        o4 = net(batches[0])
        assert type(o4.grad_fn).__name__.startswith('_GraphedPass')
        o5 = net(batches[1])                            # o4 is still referenced: eager
        assert not type(o5.grad_fn).__name__.startswith('_GraphedPass')
        del o4, o5
        o6 = net(batches[2])
        assert type(o6.grad_fn).__name__.startswith('_GraphedPass')
        o6.sum().backward()
        del o6
        # another shape, no_grad, eval: the launches one by one
        other = batch(990, N // 2)
        net.load_state_dict(ref.state_dict())           # (equal BatchNorm / dropout counters again: net has been through more forwards)
        torch.manual_seed(123)
        a = net(other)
        torch.manual_seed(123)
        with train.no_autograph():
            b = ref(other)
        assert a.grad_fn is not None and not type(a.grad_fn).__name__.startswith('_GraphedPass')
        assert_close(a, b, 2e-5, 'self-capturing model, other shape (eager path)')
        with torch.no_grad():
            assert_close(net(batches[0]), ref(batches[0]), 2e-5, 'self-capturing model, no_grad call')
        net.eval(), ref.eval()
        assert_close(net(batches[1]), ref(batches[1]), 2e-5, 'self-capturing model, eval call')
        net.train(), ref.train()
        # a deep copy has no graphs and captures for itself; a hook keeps a model eager; to() drops the graphs
        twin = copy.deepcopy(net)
        assert '_autograph' not in twin.__dict__
        o = twin(batches[0])
        assert type(o.grad_fn).__name__.startswith('_GraphedPass') and twin.__dict__['_autograph'] is not runner
        o.sum().backward()
        hooked = copy.deepcopy(net)
        h = hooked.conv1_1.register_forward_hook(lambda m, i, o_: None)
        o = hooked(batches[0])
        assert '_autograph' not in hooked.__dict__ and not type(o.grad_fn).__name__.startswith('_GraphedPass')
        h.remove()
        net.to(DEV)
        assert '_autograph' not in net.__dict__
    finally:
        train.set_autograph(False)


def test_crf_late_gradients_with_two_consumers_of_one_matrix_pair_and_a_hook():
    """(Q, P) of crf_matrices_batched feeding TWO mean-field calls, a tensor hook on Q and another use of P, under
    ops.deferred_weight_grads: the deferred dP / dQ travel out of band (the node's gradient boxes), so autograd never holds a
    tensor that the end-of-pass flush has yet to fill -- dc must equal the immediate form's, and the hook must not see garbage."""
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable
    H, n = 32, 2560
    gg = torch.Generator().manual_seed(11)
    idx = torch.randint(0, n, (1, n, 16), generator=gg)
    idx[0, :, 0] = torch.arange(n)
    tab = NeighborTable(idx.to(DEV), n)
    z1, y1, z2, y2 = (torch.randn(n, H, generator=gg).to(DEV) for _ in range(4))
    wgt = torch.linspace(0, 1, n * H, device=DEV).reshape(n, H)
    res, seen = {}, {}
    for mode in ('now', 'late'):
        c = nn.Parameter((torch.eye(H) + 0.1 * torch.randn(H, H, generator=torch.Generator().manual_seed(5))).to(DEV))
        za, ya, zb, yb = (v.clone().requires_grad_(True) for v in (z1, y1, z2, y2))

        def run():
            (mat,) = ops.crf_matrices_batched([c])
            mat[0].register_hook(lambda g, mode=mode: seen.__setitem__(mode, None if g is None else g.detach().clone()))
            o1 = ops.crf_meanfield(za, ya, c, tab, 3, k0=1, matrices=mat)
            o2 = ops.crf_meanfield(zb, yb, c, tab, 2, k0=1, matrices=mat)
            return (o1 * wgt).sum() + 0.5 * (o2 * wgt).sum() + (mat[1] * mat[1]).sum() + mat[0].sum()
        if mode == 'late':
            with ops.deferred_weight_grads():
                run().backward()
        else:
            run().backward()
        torch.cuda.synchronize()
        res[mode] = [c.grad.clone(), za.grad.clone(), ya.grad.clone(), zb.grad.clone(), yb.grad.clone()]
    for a, b, what in zip(res['now'], res['late'], ('dc', 'dz1', 'dy1', 'dz2', 'dy2')):
        assert float(a.abs().max()) > 0 and bool(torch.isfinite(b).all())
        assert_close(b, a, 1e-6, 'two consumers, late vs immediate: ' + what)
    # the hook of the late pass saw only what autograd itself carried for Q (the direct use: d sum(Q) = ones), never unfilled memory
    assert seen['late'] is not None and torch.equal(seen['late'], torch.ones(H, H, device=DEV))


def test_flat_sgd_skips_the_update_while_the_barrier_failure_word_is_set():
    """a step whose one-launch kernel timed out carries a NaN gradient; the update kernel reads the sticky failure
    word and must leave parameters AND momentum untouched (eager and captured), until ops.check_gridsync clears it."""
    from crfconv_amd import _lib, ops, optim
    from crfconv_amd.distributed import FlatGradAllReduce
    dev = torch.device('cuda', 0)
    torch.manual_seed(4)
    net = torch.nn.Linear(9, 17).to(DEV)
    bucket = FlatGradAllReduce(net)
    opt = optim.FlatSGD(bucket, lr=0.1, momentum=0.9, weight_decay=1e-4, check_every=0)
    ws = ops.gridsync_ws(dev)
    ops.check_gridsync(dev)
    word = _lib.load().crfconv_gridsync_fail_word()
    was = ops.state.small_mlp_disabled
    try:
        bucket.flat.fill_(1.0)
        opt.step()                                           # a healthy step first: momentum buffer is non-zero afterwards
        p0, b0 = opt.flat.clone(), opt.buf.clone()
        ws[word] = 0x101
        bucket.flat.fill_(float('nan'))                      # what a poisoned forward / backward leaves behind
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
        assert torch.equal(opt.flat, p0) and torch.equal(opt.buf, b0)
        with pytest.raises(_lib.CrfConvError, match='grid barrier timed out'):
            ops.check_gridsync(dev)
        bucket.flat.fill_(1.0)
        opt.step()                                           # word cleared: the update runs again, from intact state
        torch.cuda.synchronize()
        assert bool(torch.isfinite(opt.flat).all()) and not torch.equal(opt.flat, p0)
    finally:
        ops.state.small_mlp_disabled = was


def test_grid_barrier_failure_is_raised_and_disables_the_one_launch_path():
    """The sticky failure word of the grid-barrier workspace (set by a one-launch kernel whose barrier timed out) must
    reach the host: ops.check_gridsync raises, zeroes the workspace and switches the one-launch KERNEL off (the small-MLP nodes go
    on with a launch-separated forward and give the same results)."""
    from crfconv_amd import _lib, ops
    dev = torch.device('cuda', 0)
    ws = ops.gridsync_ws(dev)
    ops.check_gridsync(dev)                                 # clean: no exception
    word = _lib.load().crfconv_gridsync_fail_word()
    was = ops.state.small_mlp_disabled
    try:
        ws[word] = 0x101
        with pytest.raises(_lib.CrfConvError, match='grid barrier timed out'):
            ops.check_gridsync(dev)
        assert int(ws.abs().sum()) == 0 and ops.state.small_mlp_disabled
        ops.check_gridsync(dev)
        # the small-MLP nodes stay in use, with the forward as separate launches (product, BatchNorm) instead of the grid-barrier kernel
        assert ops._mlp_small_ok(1280, 512, 128)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(1280, 512, generator=g).to(DEV).requires_grad_(True)
        W = (torch.randn(128, 512, generator=g) / 22).to(DEV).requires_grad_(True)
        go = torch.randn(1280, 128, generator=g).to(DEV)

        def run():
            bn = nn.BatchNorm1d(128).to(DEV).train()
            x.grad = W.grad = None
            out = ops.mlp_block(x, W, bn, 0.1)
            out.backward(go)
            return out.detach().clone(), x.grad.clone(), W.grad.clone(), bn.weight.grad.clone(), bn.running_var.clone()
        off = run()
        ops.state.small_mlp_disabled = False
        on = run()
        for a, b, what in zip(off, on, ('out', 'dx', 'dW', 'dgamma', 'running_var')):
            assert_close(a, b, 1e-5, 'launch-separated forward vs one-launch kernel: ' + what)
    finally:
        ops.state.small_mlp_disabled = was


@pytest.mark.usefixtures('big_forms_from_4096')
def test_two_training_forwards_before_backward_keep_their_own_dropout_masks():
    """The fused MLP -> Dropout node keys its mask on the BatchNorm's step counter, a device word later forwards advance: the
    backward of a call must use the counter value ITS forward saw (two forwards, then one backward -- multi-view losses,
    forward-all-then-backward accumulation -- used to trip autograd's in-place check, and would have masked with the wrong counter)."""
    from crfconv_amd import ops
    g = torch.Generator().manual_seed(8)
    m, ci, co = 8192, 32, 128
    W = nn.Parameter((torch.randn(co, ci, generator=g) / 6).to(DEV))
    bn = nn.BatchNorm1d(co).to(DEV).train()
    x1 = torch.randn(m, ci, generator=g).to(DEV).requires_grad_()
    x2 = torch.randn(m, ci, generator=g).to(DEV).requires_grad_()
    r1, r2 = torch.randn(m, co, generator=g).to(DEV), torch.randn(m, co, generator=g).to(DEV)

    def reset():
        for t_ in (W, bn.weight, bn.bias, x1, x2):
            t_.grad = None
        bn.num_batches_tracked.fill_(10)
    reset()
    o1 = ops.mlp_block_dropout(x1, W, bn, 0.1, 0.5)
    assert o1 is not None
    (o1 * r1).sum().backward()                               # one after the other: counters 11, then 12
    o2 = ops.mlp_block_dropout(x2, W, bn, 0.1, 0.5)
    (o2 * r2).sum().backward()
    ref = [t_.grad.clone() for t_ in (W, bn.weight, bn.bias, x1, x2)]
    keep1, keep2 = (o1 != 0), (o2 != 0)
    reset()
    a1 = ops.mlp_block_dropout(x1, W, bn, 0.1, 0.5)          # both forwards first (the counter is at 12 when the backward runs)
    a2 = ops.mlp_block_dropout(x2, W, bn, 0.1, 0.5)
    assert torch.equal(a1 != 0, keep1) and torch.equal(a2 != 0, keep2) and not torch.equal(keep1, keep2)
    ((a1 * r1).sum() + (a2 * r2).sum()).backward()
    for got, want, name in zip((W, bn.weight, bn.bias, x1, x2), ref, ('dW', 'dgamma', 'dbeta', 'dx1', 'dx2')):
        assert_close(got.grad, want, 1e-5, name)
