"""-m gpu: kNN and grid-subsampling HIP kernels against the oracle (bit-exact) and the fixtures
captured from the reference's compiled C++ (tests/golden/g6_knn.npz, g7_grid.npz)."""
import numpy as np
import pytest
import torch

import _seeded as S
from oracle import native as onative

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def nn_mod():
    from crfconv_amd.utils import nearest_neighbors
    return nearest_neighbors


@pytest.mark.parametrize('K', [1, 16, 32])
def test_knn_golden_host_api(golden, nn_mod, K):
    g = golden('g6_knn.npz')
    a = nn_mod.knn_batch(g['pts'], g['pts'], K, omp=True)
    b = nn_mod.knn_batch(torch.from_numpy(g['pts']), g['qry'], K)
    assert a.dtype == np.int64 and a.shape == (2, 2048, K)
    assert np.array_equal(a, g['self_K%d' % K].astype(np.int64))
    assert np.array_equal(b, g['cross_K%d' % K].astype(np.int64))
    assert np.array_equal(nn_mod.knn(g['pts'][1], g['qry'][1], K), b[1])
    assert np.array_equal(nn_mod.knn(g['pts'][0], g['pts'][0], K, omp=True), a[0])


def test_knn_lattice_distance_rows(golden, nn_mod):
    g = golden('g6_knn.npz')
    lat = g['lattice_pts']
    idx = nn_mod.knn(lat, lat, 16)
    assert np.array_equal(onative.knn_sq_dists(lat, lat, idx), g['lattice_dists'])
    # ties -> (distance, index) order, which is what the oracle restates
    assert np.array_equal(idx, onative.oracle_knn(lat, lat, 16))


@pytest.mark.parametrize('B,Np,Nq,K,box', [
    (1, 16, 16, 16, (1, 1, 1)),            # K == npts
    (3, 1000, 257, 8, (1, 1, 1)),          # K between template sizes, ragged block
    (2, 5000, 5000, 16, (8, 8, 0)),        # planar cloud (degenerate z extent)
    (2, 3000, 700, 33, (60, 1, 1)),        # elongated, K > 32 path
    (1, 1, 5, 1, (1, 1, 1)),               # single support point
    (4, 40960, 40960, 16, (8, 8, 3)),      # BASELINE config-2 level-0 shape
])
def test_knn_vs_oracle(nn_mod, B, Np, Nq, K, box):
    pts = np.stack([S.make_cloud(900 + b, Np, box=box) for b in range(B)])
    if Nq == Np:
        qry = pts
    else:  # queries partly outside the support bounding box
        qry = np.stack([S.make_cloud(950 + b, Nq, box=box) * 1.3 - 0.1 for b in range(B)]).astype(np.float32)
    dev = nn_mod.knn_batch_device(torch.from_numpy(pts).cuda(), torch.from_numpy(qry).cuda(), K)
    want = onative.oracle_knn_batch(pts, qry, K)
    got = dev.cpu().numpy()
    assert got.dtype == np.int64
    assert np.array_equal(got, want)
    i32 = nn_mod.knn_batch_device(torch.from_numpy(pts).cuda(), torch.from_numpy(qry).cuda(), K, torch.int32)
    assert np.array_equal(i32.cpu().numpy().astype(np.int64), want)


def test_knn_duplicates_and_errors(nn_mod):
    base = S.make_cloud(7, 500)
    pts = np.concatenate([base, base[:100]])[None]           # exact duplicates
    got = nn_mod.knn_batch(pts, pts, 8)
    assert np.array_equal(got, onative.oracle_knn_batch(pts, pts, 8))
    from crfconv_amd._lib import CrfConvError
    with pytest.raises(CrfConvError):
        nn_mod.knn_batch(pts, pts, 700)                      # K > npts
    with pytest.raises(CrfConvError):
        nn_mod.knn_batch(np.zeros((1, 10, 2), np.float32), np.zeros((1, 10, 2), np.float32), 2)   # dim != 3


@pytest.mark.parametrize('Np,K', [(6000, 16), (6000, 8), (6000, 2), (6000, 24), (6000, 1), (3000, 16), (3000, 8), (3000, 32),
                                  (3000, 1), (4096, 16), (4097, 16), (40, 16)])
def test_knn_clustered_duplicates_every_search_path(nn_mod, Np, K):
    """The three searches of csrc/knn.hip on hostile input -- blobs of very different density, a planar sheet, 10 % exact
    duplicates (ties -> lower id), queries far outside the support -- against the oracle, bit for bit: sixteen lanes per
    query over the whole cloud (<= 4096 points, K in {1, 8, 16, 32}), sixteen lanes per query on the grid with the LDS
    rank selection (larger clouds, 1 < K <= 16; the dense blob overflows the candidate pool and forces mid-ring
    compactions), one lane per query (everything else)."""
    rng = np.random.default_rng(Np * 131 + K)
    parts = [rng.normal(0, 0.01, (Np // 4, 3)), rng.normal(2, 0.5, (Np // 4, 3)),
             np.concatenate([rng.uniform(-3, 3, (Np // 4, 2)), np.full((Np // 4, 1), 0.25)], 1)]
    rest = Np - 3 * (Np // 4)
    n_dup = rest // 2
    parts.append(rng.uniform(-4, 4, (rest - n_dup, 3)))
    pts = np.concatenate(parts).astype(np.float32)
    pts = np.concatenate([pts, pts[rng.integers(0, len(pts), n_dup)]])[rng.permutation(Np)]
    assert pts.shape == (Np, 3)
    pts = np.stack([pts, pts[::-1] * np.float32(1.5)])                         # two clouds, different scales
    qry = np.concatenate([pts[:, : Np // 2], pts[:, : Np // 4] * np.float32(3.0) + np.float32(7.0)], 1)
    for q in (pts, qry):
        got = nn_mod.knn_batch_device(torch.from_numpy(pts).cuda(), torch.from_numpy(np.ascontiguousarray(q)).cuda(), K)
        assert np.array_equal(got.cpu().numpy(), onative.oracle_knn_batch(pts, np.ascontiguousarray(q), K))


@pytest.mark.parametrize('K', [32, 16])
def test_knn_large_properties(nn_mod, K):
    """~1M-point scene (BASELINE config 5 scale): size-independent properties + a sampled brute-force check on the
    device.  K = 32 runs the one-lane grid search, K = 16 the sixteen-lanes-per-query one."""
    n = 1 << 20
    g = torch.Generator().manual_seed(5)
    pts = (torch.rand(1, n, 3, generator=g) * torch.tensor([60.0, 60.0, 15.0])).cuda()
    idx = nn_mod.knn_batch_device(pts, pts, K)
    p = pts[0]
    nb = p[idx[0]]                                           # [n, K, 3]
    d = ((p[:, None, :] - nb) ** 2).sum(-1)
    assert bool((idx[0, :, 0] == torch.arange(n, device='cuda')).all())      # self first
    assert bool((d[:, 1:] >= d[:, :-1] - 1e-6).all())                        # ascending
    rows = torch.randint(0, n, (256,), generator=g).cuda()
    full = ((p[rows][:, None, :] - p[None, :, :]) ** 2).sum(-1)              # [256, n]
    kth = full.topk(K, largest=False).values[:, -1]
    assert torch.allclose(d[rows, -1], kth, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ grid subsampling
def _rekey_rows(pts_in, dl, rows):
    op, _, _, okeys = onative.oracle_grid_subsample(pts_in, None, None, dl)
    lut = {tuple(p): k for p, k in zip(map(tuple, op), okeys)}
    return np.array([lut[tuple(p)] for p in map(tuple, rows)], dtype=np.uint64)


@pytest.mark.parametrize('name', ['all', 'two', 'ponly', 'fonly', 'conly'])
def test_grid_golden(golden, name):
    from crfconv_amd.utils import cpp_subsampling
    g = golden('g7_grid.npz')
    dl = float(g[name + '/dl'])
    kw = {}
    if name in ('all', 'two', 'fonly'):
        kw['features'] = g['feats']
    if name in ('all', 'conly'):
        kw['classes'] = g['lab1']
    if name == 'two':
        kw['classes'] = g['lab2']
    res = cpp_subsampling.compute(g['pts'], sampleDl=dl, **kw)
    want = onative.oracle_grid_subsample(g['pts'], kw.get('features'), kw.get('classes'), dl)
    if name == 'ponly':
        assert isinstance(res, np.ndarray)
        res = (res,)
    else:
        assert isinstance(res, tuple) and len(res) == 1 + len(kw)
    assert np.array_equal(res[0], want[0])                   # bit-exact vs the oracle, same row order
    order = np.argsort(_rekey_rows(g['pts'], dl, g[name + '/pts']), kind='stable')
    assert np.array_equal(res[0], g[name + '/pts'][order])   # bit-exact vs the reference, re-keyed
    i = 1
    if 'features' in kw:
        assert res[i].dtype == np.float32
        assert np.array_equal(res[i], want[1])
        assert np.array_equal(res[i], g[name + '/feats'][order])
        i += 1
    if 'classes' in kw:
        assert res[i].dtype == np.int32 and res[i].ndim == 2
        assert np.array_equal(res[i], want[2])


def test_grid_errors_and_shapes():
    from crfconv_amd.utils import cpp_subsampling
    pts = S.make_cloud(3, 100)
    with pytest.raises(RuntimeError, match='points.shape is not'):
        cpp_subsampling.compute(pts[:, :2])
    with pytest.raises(RuntimeError, match='features.shape is not'):
        cpp_subsampling.compute(pts, features=np.zeros((99, 3), np.float32))
    with pytest.raises(RuntimeError, match='Error parsing method'):
        cpp_subsampling.compute(pts, method='nope')
    with pytest.raises(TypeError):
        cpp_subsampling.compute(pts, np.zeros((100, 3), np.float32))      # keyword-only
    one = cpp_subsampling.compute(pts, sampleDl=10.0)                      # everything in one voxel
    assert one.shape == (1, 3)


def test_grid_large_vs_oracle():
    from crfconv_amd.utils import cpp_subsampling
    n = 2_000_000
    pts = S.make_cloud(11, n, box=(30.0, 30.0, 8.0))
    feats = S.uniform(11, 'f', (n, 3), 0, 255)
    lab = S.integers(11, 'l', (n,), 0, 9).astype(np.int32)
    p, f, c = cpp_subsampling.compute(pts, features=feats, classes=lab, sampleDl=0.06)
    wp, wf, wc, _ = onative.oracle_grid_subsample(pts, feats, lab, 0.06)
    assert np.array_equal(p, wp) and np.array_equal(f, wf) and np.array_equal(c, wc)


def test_knn_batch_from_forked_dataloader_style_worker():
    """SURVEY 8(b) threading row: importing the package creates no HIP context, so a fork()ed worker process can call the
    kNN extension (the reference calls it inside the DataLoader collate function, datasets/semantic3d_dataset.py:498).  The
    parent must not have initialised the GPU, so the parent side runs in a fresh interpreter: it imports crfconv_amd, forks
    two workers, and each worker's first kNN call creates its own context."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import multiprocessing as mp, sys
sys.path.insert(0, %r)
import numpy as np
import crfconv_amd                                   # import only: no HIP call in the parent
from crfconv_amd.utils import nearest_neighbors

def worker(q):
    rng = np.random.default_rng(7)
    pts = rng.random((2, 500, 3)).astype(np.float32)
    idx = nearest_neighbors.knn_batch(pts, pts, 8, omp=True)
    d = ((pts[:, :, None, :] - pts[:, None, :, :]) ** 2).sum(-1)
    q.put(bool(np.array_equal(np.sort(idx, -1), np.sort(np.argsort(d, -1, kind='stable')[:, :, :8], -1))))

ctx = mp.get_context('fork'); q = ctx.Queue()
ps = [ctx.Process(target=worker, args=(q,)) for _ in range(2)]
[p.start() for p in ps]; res = [q.get(timeout=120) for _ in ps]; [p.join() for p in ps]
assert all(res) and all(p.exitcode == 0 for p in ps), (res, [p.exitcode for p in ps])
print('fork ok')
""" % root
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'fork ok' in out.stdout, out.stderr[-2000:]


def test_morton_codes_kernel_equals_the_op_by_op_form():
    """crfconv_morton_codes (two launches) against the framework-op form it replaced (data.morton_codes on a host copy): the
    same codes bit for bit -- random clouds, a degenerate cloud (all points equal), a planar one, negative coordinates."""
    from crfconv_amd import data
    g = torch.Generator().manual_seed(21)
    clouds = [torch.rand(3, 5000, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0]) - torch.tensor([4.0, 1.0, 0.5]),
              torch.ones(2, 257, 3) * 0.37,
              torch.cat([torch.rand(2, 1000, 2, generator=g), torch.zeros(2, 1000, 1)], -1),
              torch.randn(1, 40960, 3, generator=g) * 100.0]
    for pos in clouds:
        ref = data.morton_codes(pos)                                   # host tensor: op-by-op path
        got = data.morton_codes(pos.to(DEV))
        assert got.dtype == torch.int64 and torch.equal(got.cpu(), ref)
        assert torch.equal(data.morton_order(pos.to(DEV)).cpu(), torch.argsort(ref, dim=1, stable=True))



def test_reverse_csr_hub_rows():
    """Rows of very large in-degree (every target also points at row 0 / row 1 of its cloud) take the tiled ranking path of
    rev_sort_rows_kernel: the reverse lists must still be every row's incoming edge ids in ascending order."""
    from crfconv_amd.graph import NeighborTable
    g = torch.Generator().manual_seed(9)
    B, N, K = 2, 5000, 16
    idx = torch.randint(0, N, (B, N, K), generator=g)
    idx[:, :, 1] = 0
    idx[:, ::2, 2] = 1
    tab = NeighborTable(idx.to('cuda'), N)
    rev_ptr, rev_eid = (v.cpu().long() for v in tab.reverse)
    src = tab.idx32.cpu().long().reshape(-1)
    order = torch.argsort(src, stable=True)                 # stable sort by source row = ascending edge id inside a row
    assert torch.equal(rev_eid, order)
    cnt = torch.bincount(src, minlength=B * N)
    assert torch.equal(rev_ptr, torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]))
    assert int(cnt.max()) >= N


def test_reverse_csr_degenerate_cloud_one_hub_of_every_edge():
    """All 40 960 x 16 entries of a table name the same source (a cloud of coincident points): ONE reverse row of 655 360 edge ids.
    The wavefront-wide radix sort of graph.hip handles it in linear time (the all-pairs ranking would be ~1e9 tile comparisons),
    through the single-table build and the batched one; both must give the ascending edge ids."""
    import time
    from crfconv_amd.graph import NeighborTable, batched_reverse
    N, K = 40960, 16
    idx = torch.full((1, N, K), 7, dtype=torch.int64)
    idx[0, 123, 5] = 9                                        # and one ordinary row
    tab = NeighborTable(idx.to('cuda'), N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rev_ptr, rev_eid = tab.reverse
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    src = tab.idx32.cpu().long().reshape(-1)
    want = torch.argsort(src, stable=True)
    assert torch.equal(rev_eid.cpu().long(), want)
    assert int(rev_ptr[8] - rev_ptr[7]) == N * K - 1 and int(rev_ptr[10] - rev_ptr[9]) == 1
    assert dt < 0.5, 'hub row took %.3f s' % dt
    rev_eid.fill_(-1)
    with batched_reverse():
        tab._build_reverse(*tab._rev)
    assert torch.equal(tab._rev[1].cpu().long(), want)


def test_argsort_codes_equals_torch_stable_argsort():
    """crfconv_argsort_codes (bucket by the top 16 bits, rank inside the bucket by (code, index)) against
    torch.argsort(stable=True): random 30-bit codes, heavy ties, one value only, and real Morton codes."""
    from crfconv_amd.data import morton_codes, morton_order
    from crfconv_amd import _lib
    from crfconv_amd.graph import ptr, stream_ptr
    g = torch.Generator().manual_seed(4)

    def run(code):
        B, N = code.shape
        order = torch.empty_like(code)
        nbytes = _lib.load().crfconv_argsort_codes_workspace(B, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=code.device)
        _lib.call('crfconv_argsort_codes', ptr(code), B, N, ptr(order), ptr(ws), nbytes, stream_ptr())
        return order
    for code in (torch.randint(0, 1 << 30, (3, 5000), generator=g), torch.randint(0, 7, (2, 4096), generator=g) << 20,
                 torch.full((1, 3000), 12345), torch.randint(0, 1 << 30, (1, 1), generator=g),
                 torch.randint(0, 1 << 14, (2, 9000), generator=g)):
        code = code.to('cuda')
        assert torch.equal(run(code), torch.argsort(code, dim=1, stable=True))
    pos = torch.rand(4, 40960, 3, generator=g).to('cuda')
    assert torch.equal(morton_order(pos), torch.argsort(morton_codes(pos), dim=1, stable=True))
    # degenerate clouds at full size: every point in ONE bucket -- the wavefront-cooperative hub ranking, linear
    # loads per element instead of N dependent load pairs; and hub + ordinary buckets inside one wavefront, ragged tail
    dup = torch.full((4, 40960), 777 << 14).to('cuda')
    assert torch.equal(run(dup), torch.arange(40960, device='cuda').expand(4, -1))
    mixed = torch.randint(0, 1 << 30, (2, 1003), generator=g)
    mixed[:, 100:480] = 5 << 14
    mixed[1, 700:] = (9 << 14) + 3
    mixed = mixed.to('cuda')
    assert torch.equal(run(mixed), torch.argsort(mixed, dim=1, stable=True))
    near = (torch.randint(0, 3, (2, 40960), generator=g) + (4242 << 14)).to('cuda')      # three distinct codes, one bucket
    assert torch.equal(run(near), torch.argsort(near, dim=1, stable=True))


def test_random_subsets_device():
    """crfconv_random_subsets: exact sizes, ascending, distinct, in range; a function of (seed, counter, level); new subsets
    when the counter advances; every point is drawn about equally often."""
    from crfconv_amd.data import random_subsets_device
    sizes, counts = [40960, 10240, 2560, 640, 160, 7], [10240, 2560, 640, 160, 80, 7]
    ctr = torch.zeros(1, dtype=torch.int64, device='cuda')

    def draw(seed):
        outs = [torch.full((c,), -1, dtype=torch.int64, device='cuda') for c in counts]
        random_subsets_device(sizes, counts, seed, ctr, outs)
        return outs
    a, a2, b = draw(11), draw(11), draw(12)
    for n, c, t_ in zip(sizes, counts, a):
        v = t_.cpu()
        assert v.numel() == c and int(v.min()) >= 0 and int(v.max()) < n
        assert bool((v[1:] > v[:-1]).all())                  # ascending and distinct
    assert all(torch.equal(u, v) for u, v in zip(a, a2))      # reproducible
    assert not torch.equal(a[0], b[0])                        # another seed, another subset
    assert not torch.equal(a[1][:640], a[2])                  # levels draw independently
    ctr += 1
    c = draw(11)
    assert not torch.equal(a[0], c[0])                        # the counter advances the stream
    hits = torch.zeros(2560, dtype=torch.int64)
    for it in range(200):                                     # uniformity: each of 2560 points is in a quarter of the draws
        ctr += 1
        hits += torch.bincount(draw(5)[2].cpu(), minlength=2560)
    assert abs(float(hits.float().mean()) - 50.0) < 1e-6 and int(hits.min()) >= 20 and int(hits.max()) <= 85


def test_up_index_from_the_neighbour_table_equals_the_k1_search():
    """data.up_index_from_table (round 5: the nearest subset member of a point read off its own K-nearest table, a wavefront-wide
    scan of the subset for the points without a member in the table) must be bit-identical to knn_batch(sub_pos, pos, 1)
    (datasets/semantic3d_dataset.py:524): random clouds, an UNSORTED subset (ties go to the lower subset position), tie-heavy lattice
    coordinates, a short table, a subset so thin that most points take the scan, the membership table of the device draw."""
    from crfconv_amd.data import random_subsets_device, up_index_from_table
    from crfconv_amd.utils.nearest_neighbors import knn_batch_device
    g = torch.Generator().manual_seed(77)

    def check(pos, choice, K, rank=None, what=''):
        nbr = knn_batch_device(pos, pos, K)
        want = knn_batch_device(pos[:, choice].contiguous(), pos, 1)
        got = up_index_from_table(pos, nbr, choice, rank)
        assert got.shape == want.shape and got.dtype == torch.int64
        assert torch.equal(got, want), (what, int((got != want).sum()))
    for B, N, ratio, K in ((4, 40960, 4, 16), (2, 10240, 4, 16), (3, 2560, 2, 16), (2, 4096, 64, 16), (2, 4096, 4, 4), (1, 777, 3, 16), (2, 100, 4, 32)):
        pos = torch.rand(B, N, 3, generator=g).to('cuda')
        choice = torch.randperm(N, generator=g)[: N // ratio].to('cuda')
        check(pos, choice, K, what='unsorted %s' % ((B, N, ratio, K),))
        check(pos, choice.sort().values, K, what='sorted %s' % ((B, N, ratio, K),))
    lat = (torch.randint(0, 32, (2, 8192, 3), generator=g).float() / 32).to('cuda')          # many equal distances, duplicate points
    for ratio in (4, 16):
        choice = torch.randperm(8192, generator=g)[: 8192 // ratio].to('cuda')
        check(lat, choice, 16, what='lattice unsorted %d' % ratio)
        check(lat, choice.sort().values, 16, what='lattice sorted %d' % ratio)
    # the membership tables the device draw leaves
    sizes, counts = [40960, 10240], [10240, 2560]
    ctr = torch.ones(1, dtype=torch.int64, device='cuda')
    outs = [torch.empty(c, dtype=torch.int64, device='cuda') for c in counts]
    ranks = [torch.full((n,), -7, dtype=torch.int32, device='cuda') for n in sizes]
    random_subsets_device(sizes, counts, 5, ctr, outs, ranks=ranks)
    for n, c, o, r in zip(sizes, counts, outs, ranks):
        ref = torch.full((n,), -1, dtype=torch.int32, device='cuda')
        ref[o] = torch.arange(c, dtype=torch.int32, device='cuda')
        assert torch.equal(r, ref)
    pos = torch.rand(4, 40960, 3, generator=g).to('cuda')
    check(pos, outs[0], 16, rank=ranks[0], what='device draw')


def test_reverse_csr_batched_equals_one_by_one():
    """graph.batched_reverse: the reverse CSRs of several tables of different shapes (a hub row, K = 1, K = 32)
    built by one crfconv_reverse_csr_batched call must equal the one-table-at-a-time builds bit for bit."""
    from crfconv_amd.graph import NeighborTable, batched_reverse
    g = torch.Generator().manual_seed(21)
    specs = [(2, 3000, 3000, 16), (2, 700, 3000, 16), (2, 3000, 700, 1), (1, 64, 64, 32), (3, 1000, 1000, 16)]
    tabs = []
    for B, n_tgt, n_src, K in specs:
        idx = torch.randint(0, n_src, (B, n_tgt, K), generator=g)
        if K == 16 and n_tgt == 1000:
            idx[:, :, 3] = 5                                  # a hub
        tabs.append(NeighborTable(idx.to('cuda'), n_src))
    ref = [tuple(v.clone() for v in t.reverse) for t in tabs]
    for t in tabs:
        for v in t._rev:
            v.fill_(-7)
    with batched_reverse():
        for t in tabs:
            t._build_reverse(*t._rev)
    for t, (rp, re) in zip(tabs, ref):
        assert torch.equal(t._rev[0], rp) and torch.equal(t._rev[1], re)


def test_pick_rows_batched_gather_equals_framework_indexing():
    """data.pick_rows (crfconv_gather_rows_batched): shared and per-cloud picks of several tensors in one launch."""
    from crfconv_amd.data import pick_rows
    g = torch.Generator().manual_seed(5)
    B, N, S = 3, 1000, 257
    pos = torch.randn(B, N, 3, generator=g).to(DEV)
    x = torch.randn(B, N, 6, generator=g).to(DEV)
    y = torch.randint(0, 13, (B, N), generator=g).to(DEV)                 # int64 [B, N]: 8-byte rows
    nb = torch.randint(0, N, (B, N, 16), generator=g).to(DEV)
    choice = torch.randperm(N, generator=g)[:S].sort().values.to(DEV)
    a = pick_rows([pos, None, nb], choice, per_cloud=False)
    assert a[1] is None and torch.equal(a[0], pos[:, choice]) and torch.equal(a[2], nb[:, choice])
    order = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(DEV)
    p, xx, yy, none = pick_rows([pos, x, y, None], order, per_cloud=True)
    assert none is None
    assert torch.equal(p, torch.gather(pos, 1, order[..., None].expand(-1, -1, 3)))
    assert torch.equal(xx, torch.gather(x, 1, order[..., None].expand(-1, -1, 6)))
    assert torch.equal(yy, torch.gather(y, 1, order))
    # rows that are not a whole number of dwords: the framework path
    h = torch.randint(0, 100, (B, N, 3), generator=g, dtype=torch.int16).to(DEV)
    assert torch.equal(pick_rows([h], choice, per_cloud=False)[0], h[:, choice])


def test_batched_refresh_equals_table_by_table():
    """graph.batched_reverse around NeighborTable.refresh_: narrowed indices (int32 / uint16), reverse CSRs and the memoised
    rel-pos moments written by the three batched calls must equal the one-table-at-a-time refresh bit for bit, bad entries are
    counted (and clamped) the same way, and a locality-free table (every source far from its target: the LDS window of the
    reverse-CSR passes misses) takes the fallback atomics."""
    from crfconv_amd import ops
    from crfconv_amd.graph import NeighborTable, batched_reverse
    g = torch.Generator().manual_seed(33)
    specs = [(2, 3000, 3000, 16), (2, 700, 3000, 16), (2, 3000, 700, 1), (1, 64, 64, 32), (2, 5000, 5000, 16)]

    def draw(B, n_tgt, n_src, K, local):
        if local:                                             # Morton-like: sources near the target's own row
            base = (torch.arange(n_tgt) * n_src // n_tgt)[None, :, None]
            return (base + torch.randint(-40, 40, (B, n_tgt, K), generator=g)).clamp_(0, n_src - 1)
        return torch.randint(0, n_src, (B, n_tgt, K), generator=g)

    def build(batched, bad):
        gs = torch.Generator().manual_seed(7)
        tabs, poss = [], []
        for i, (B, n_tgt, n_src, K) in enumerate(specs):
            t = NeighborTable(draw(B, n_tgt, n_src, K, i == 4).to(DEV), n_src)
            t.reverse
            if K == 16:
                ps, pt = torch.randn(B * n_src, 3, generator=gs).to(DEV), torch.randn(B * n_tgt, 3, generator=gs).to(DEV)
                t.cache['m'] = ops.MomentsEntry(ops.relpos_moments(ps, pt, t), ps, pt)
            tabs.append(t)
        new = [draw(*s, i == 4) for i, s in enumerate(specs)]
        if bad:
            new[1][0, 5, 2] = 10 ** 6
            new[1][1, 9, 0] = -3
        if batched:
            with batched_reverse():
                for t, idx in zip(tabs, new):
                    t.refresh_(idx.to(DEV), check=False)
        else:
            for t, idx in zip(tabs, new):
                t.refresh_(idx.to(DEV), check=False)
        torch.cuda.synchronize()
        return tabs

    for bad in (False, True):
        g.manual_seed(33)
        a = build(False, bad)
        g.manual_seed(33)
        b = build(True, bad)
        for ta, tb in zip(a, b):
            assert torch.equal(ta.idx32, tb.idx32)
            assert (ta.idx16 is None) == (tb.idx16 is None) and (ta.idx16 is None or torch.equal(ta.idx16, tb.idx16))
            assert torch.equal(ta._rev[0], tb._rev[0]) and torch.equal(ta._rev[1], tb._rev[1])
            assert int(ta._bad) == int(tb._bad)
            if 'm' in ta.cache:
                for u, v in zip(ta.cache['m'], tb.cache['m']):
                    assert (u == v) if not torch.is_tensor(u) else torch.equal(u, v)
        assert int(a[1]._bad) == (2 if bad else 0)
        if bad:
            with pytest.raises(IndexError):
                b[1].validate()
            b[1].validate()                                   # the count was reset by the raise


@pytest.mark.parametrize('n32,n64', [(7, 3), (100, 5), (4, 40), (0, 2), (3, 0)])
def test_reduce_jobs_both_equals_the_two_launches(n32, n64):
    """crfconv_reduce_jobs_both: the float weight-gradient sums and the float64 sums of a backward pass in ONE launch (both tables
    fit: <= 96 / <= 32 jobs) or as the two launches (more jobs than a table holds, or one kind absent) -- bit-identical to
    crfconv_reduce_jobs + crfconv_reduce_jobs_f64 either way."""
    import ctypes
    from crfconv_amd import _lib
    from crfconv_amd.graph import stream_ptr
    g = torch.Generator().manual_seed(n32 * 100 + n64)
    p32, p64, shapes32, shapes64 = [], [], [], []
    for j in range(n32):
        nblk, nslots = int(torch.randint(1, 70, (1,), generator=g)), int(torch.randint(1, 700, (1,), generator=g))
        p32.append(torch.randn(nblk, nslots, generator=g).to(DEV))
    for j in range(n64):
        nblk, nslots = int(torch.randint(1, 200, (1,), generator=g)), int(torch.randint(1, 40, (1,), generator=g))
        isf = j % 2 == 0
        p64.append(torch.randn(nblk, nslots, generator=g, dtype=torch.float32 if isf else torch.float64).to(DEV))
    res = {}
    for mode in ('both', 'two'):
        o32 = [torch.full((p.shape[1],), float('nan'), device=DEV) for p in p32]
        o64 = [torch.full((p.shape[1],), float('nan'), device=DEV, dtype=torch.float64) for p in p64]
        a32 = (_lib.ReduceJob * max(n32, 1))(*[_lib.ReduceJob(p.data_ptr(), o.data_ptr(), p.shape[0], p.shape[1]) for p, o in zip(p32, o32)])
        a64 = (_lib.Reduce64Job * max(n64, 1))(*[_lib.Reduce64Job(p.data_ptr(), 1 if p.dtype == torch.float32 else 0, p.shape[0], p.shape[1], o.data_ptr())
                                                  for p, o in zip(p64, o64)])
        v32 = ctypes.cast(a32, ctypes.c_void_p) if n32 else None
        v64 = ctypes.cast(a64, ctypes.c_void_p) if n64 else None
        if mode == 'both':
            _lib.call('crfconv_reduce_jobs_both', v32, n32, v64, n64, stream_ptr())
        else:
            if n32:
                _lib.call('crfconv_reduce_jobs', v32, n32, stream_ptr())
            if n64:
                _lib.call('crfconv_reduce_jobs_f64', v64, n64, stream_ptr())
        torch.cuda.synchronize()
        res[mode] = o32 + o64
    for a, b in zip(res['both'], res['two']):
        assert torch.equal(a, b) and not bool(torch.isnan(a).any())
    for p, o in zip(p32 + p64, res['both']):
        want = p.double().sum(0)
        assert float((o.double() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


def test_loss_forward_fold_with_and_without_ticket_words():
    """crfconv_softmax_ce_forward: with ticket words the forward's last workgroup folds the per-block sums, without them a second
    launch does (ce_finalize_kernel) -- same arithmetic, same order: sums and loss bit for bit."""
    from crfconv_amd import _lib, ops
    from crfconv_amd.graph import ptr, stream_ptr
    g = torch.Generator().manual_seed(4)
    m, C = 70001, 13
    z = torch.randn(m, C, generator=g).to(DEV)
    tgt = torch.randint(-1, C, (m,), generator=g).to(DEV)
    w = (0.5 + torch.rand(C, generator=g)).to(DEV)
    out = {}
    for name, ticket in (('ticket', ops._ticket(DEV)), ('launch', None)):
        lse = torch.empty(m, device=DEV)
        sums = torch.empty(3, dtype=torch.float64, device=DEV)
        loss = torch.empty((), device=DEV)
        nbytes = _lib.load().crfconv_softmax_ce_workspace(m)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        for _ in range(2):                                         # twice: the ticket words must be left zero
            _lib.call('crfconv_softmax_ce_forward', ptr(z), ptr(tgt), ptr(w), m, C, -1, 0, ptr(lse), ptr(sums), ptr(loss), ptr(ws), nbytes,
                      ptr(ticket), stream_ptr())
        torch.cuda.synchronize()
        out[name] = (lse, sums, loss)
    for a, b in zip(out['ticket'], out['launch']):
        assert torch.equal(a, b)
    ref = torch.nn.functional.cross_entropy(z.double(), tgt, weight=w.double(), ignore_index=-1)
    assert abs(float(out['ticket'][2]) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    assert int(ops._ticket(DEV).abs().sum()) == 0


def test_uv_hosting_refuses_other_widths():
    """crfconv_pointconv_forward_uv_hosting is for the widths crfconv_pointconv_forward_uv_hosts() names (d = 8): another width is an
    error of the call, never a silently different launch."""
    from crfconv_amd import _lib
    lib = _lib.load()
    assert lib.crfconv_pointconv_forward_uv_hosts(16, 8) == 1 and lib.crfconv_pointconv_forward_uv_hosts(16, 16) == 0
    assert lib.crfconv_pointconv_forward_uv_hosts(16, 64) == 0
    dummy = torch.zeros(64, device=DEV)
    with pytest.raises(_lib.CrfConvError):
        _lib.call('crfconv_pointconv_forward_uv_hosting', dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), 16, 4, 16,
                  dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), 0.1, dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(),
                  dummy.data_ptr(), dummy.data_ptr(), 64, None, None, None, 1, None, None, None)
