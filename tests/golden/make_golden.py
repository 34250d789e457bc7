#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz by RUNNING THE REFERENCE (this container only).

    python tests/golden/make_golden.py          # needs /root/reference and `make -C oracle ref`

What runs is the reference's own code:
  * models/  (continuous_crf_conv_big.py, point_conv_big.py, common.py, continuous_crf_conv.py,
    point_conv.py) imported from /root/reference with sys.modules stubs for the third-party
    packages that are absent everywhere (torch_geometric, torch_scatter, torch_points3d).  The
    stubs are minimal restatements of those packages' documented semantics (marked below);
    results that pass through them are "parity unpinned" at that boundary.
  * utils/nearest_neighbors/knn_.cxx and utils/cpp_wrappers/.../grid_subsampling.cpp compiled
    unchanged into oracle/_ref/ (oracle/Makefile) and called through ctypes.
Only data (inputs, parameters, outputs, gradients) is written; no reference source travels.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('CRFCONV_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _seeded as S  # noqa: E402
from oracle import native as onative  # noqa: E402


# ----------------------------------------------------------------------------- third-party stubs
class FastBatchNorm1d(nn.Module):
    """torch_points3d.core.common_modules.FastBatchNorm1d restated (upstream: BatchNorm1d held as
    .batch_norm, momentum 0.1; 3-D input normalised over [B, N] by transposing to [B, C, N])."""

    def __init__(self, num_features, momentum=0.1, **kw):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(num_features, momentum=momentum, **kw)

    def forward(self, x):
        if x.dim() == 2:
            return self.batch_norm(x)
        if x.dim() == 3:
            return self.batch_norm(x.transpose(1, 2)).transpose(2, 1)
        raise ValueError('Non supported number of dimensions {}'.format(x.dim()))


def _pyg_softmax(src, index, ptr=None, num_nodes=None):
    """torch_geometric.utils.softmax restated: per-target-node softmax (max-shifted, +1e-16)."""
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    mx = torch.full((n,) + src.shape[1:], float('-inf'), dtype=src.dtype)
    mx = mx.scatter_reduce(0, index.view(-1, *[1] * (src.dim() - 1)).expand_as(src), src, 'amax')
    out = (src - mx[index]).exp()
    den = torch.zeros((n,) + src.shape[1:], dtype=src.dtype).index_add_(0, index, out)
    return out / (den[index] + 1e-16)


def _scatter_add(src, index, dim=0, out=None, dim_size=None):
    """torch_scatter.scatter_add restated (dim 0 only)."""
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    return torch.zeros((n,) + src.shape[1:], dtype=src.dtype).index_add_(0, index, src)


def _scatter_max(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    res = torch.full((n,) + src.shape[1:], float('-inf'), dtype=src.dtype)
    res = res.scatter_reduce(0, index.view(-1, 1).expand_as(src), src, 'amax')
    return res, None


def _remove_self_loops(edge_index, edge_attr=None):
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], None


def _add_self_loops(edge_index, edge_weight=None, fill_value=1, num_nodes=None):
    loops = torch.arange(num_nodes, dtype=edge_index.dtype)
    return torch.cat([edge_index, torch.stack([loops, loops])], 1), None


class _MessagePassing(nn.Module):
    """torch_geometric.nn.MessagePassing restated for the one use in the reference
    (aggr='add', flow='source_to_target': j = edge_index[0] sends to i = edge_index[1])."""

    def propagate(self, edge_index, x, pos):
        col, row = edge_index
        if torch.is_tensor(pos):
            pos_j, pos_i, n = pos[col], pos[row], pos.shape[0]
        else:
            pos_j, pos_i, n = pos[0][col], pos[1][row], pos[1].shape[0]
        msg = self.message(x_j=x[col], pos_i=pos_i, pos_j=pos_j)
        return torch.zeros((n, msg.shape[1]), dtype=msg.dtype).index_add_(0, row, msg)


_INJECTED_GRAPH = {}


def _radius_graph(pos, r, batch=None, loop=False, max_num_neighbors=32):
    """Stand-in: the graph itself is supplied by the caller (torch_cluster's radius search is not
    available); returns the injected [2, E] (col = source, row = target)."""
    return _INJECTED_GRAPH['edge_index']


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    na = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError('third-party op not available'))
    inits = mod('torch_geometric.nn.inits', zeros=na, glorot=na, reset=na)
    tgnn = mod('torch_geometric.nn', MessagePassing=_MessagePassing, fps=na, radius=na, knn=na,
               radius_graph=_radius_graph, knn_graph=na, GMMConv=na, PointConv=na,
               knn_interpolate=na, inits=inits)
    tgu = mod('torch_geometric.utils', softmax=_pyg_softmax, remove_self_loops=_remove_self_loops,
              add_self_loops=_add_self_loops)
    mod('torch_geometric', nn=tgnn, utils=tgu)
    mod('torch_scatter', scatter=na, scatter_add=_scatter_add, scatter_max=_scatter_max)
    cm = mod('torch_points3d.core.common_modules', FastBatchNorm1d=FastBatchNorm1d)
    core = mod('torch_points3d.core', common_modules=cm)
    mod('torch_points3d', core=core)


def import_reference_models():
    install_stubs()
    sys.path.insert(0, REF)
    import models  # noqa: F401  (the reference package)
    return models


# ----------------------------------------------------------------------------- helpers
def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_seeded(module, seed):
    sd = S.fill_state_dict(shapes_of(module), seed)
    module.load_state_dict(sd, strict=True)
    return sd


def grads_of(module):
    return {k: p.grad.detach().clone() for k, p in module.named_parameters()}


def buffers_after(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()
            if 'running_' in k or 'num_batches' in k}


def pack(prefix, d):
    return {'%s/%s' % (prefix, k): (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


def ref_knn(support, query, k):
    return onative.ref_knn_batch(support, query, k, omp=False)


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# ----------------------------------------------------------------------------- G1/G2: dense CRFConv
def g1_crfconv(models):
    from models.continuous_crf_conv_big import ContinuousGaussianCRFConv
    B, N, Nc, K = 2, 256, 64, 16
    U, P, O = 64, 32, 32
    out = {}
    pos = np.stack([S.make_cloud(100 + b, N) for b in range(B)])
    choice = S.permutation(1, 'g1choice', N)[:Nc]
    nbr = ref_knn(pos, pos, K)
    up = ref_knn(np.ascontiguousarray(pos[:, choice]), pos, 1)
    unary = torch.from_numpy(S.uniform(1, 'g1unary', (B, Nc, U)))
    pairwise = torch.from_numpy(S.uniform(1, 'g1pair', (B, N, P)))
    gout = torch.from_numpy(S.uniform(1, 'g1gout', (B, N, O)))
    out.update(pos=pos, neighbor_idx=nbr.astype(np.int16), up_idx=up.astype(np.int16),
               unary=unary.numpy(), pairwise=pairwise.numpy(), gout=gout.numpy())
    for steps, mode in ((1, 'train'), (3, 'train'), (3, 'eval'), (5, 'eval')):
        if True:
            m = ContinuousGaussianCRFConv(U, P, O, steps=steps)
            sd = load_seeded(m, 7)
            m.train(mode == 'train')
            u = unary.clone().requires_grad_(True)
            p = pairwise.clone().requires_grad_(True)
            y = m(u, p, torch.from_numpy(up), torch.from_numpy(nbr))
            (y * gout).sum().backward()
            tag = 'T%d_%s' % (steps, mode)
            out[tag + '/out'] = y.detach().numpy()
            out[tag + '/d_unary'] = u.grad.numpy()
            out[tag + '/d_pairwise'] = p.grad.numpy()
            out.update(pack(tag + '/grad', grads_of(m)))
            if mode == 'train':
                out.update(pack(tag + '/buf', buffers_after(m)))
    out.update(pack('sd', sd))
    save('g1_crfconv.npz', **out)


def g2_meanfield_fp64(models):
    """The mean-field core alone in float64 (high-precision anchor), via the reference module's
    own _compute_similarity / loop run on double tensors."""
    from models.continuous_crf_conv_big import ContinuousGaussianCRFConv as M
    B, N, K, H = 2, 256, 16, 8
    pos = np.stack([S.make_cloud(200 + b, N) for b in range(B)])
    nbr = torch.from_numpy(ref_knn(pos, pos, K))[:, :, 1:]
    z = torch.from_numpy(S.uniform(2, 'g2z', (B, N, H))).double()
    y = torch.from_numpy(S.uniform(2, 'g2y', (B, N, H))).double()
    c = (torch.eye(H) + 0.1 * torch.from_numpy(S.uniform(2, 'g2c', (H, H)))).double()
    out = dict(z=z.numpy(), y=y.numpy(), c=c.numpy(), nbr=nbr.numpy().astype(np.int16))
    holder = Obj()
    s = M._compute_similarity(holder, y, nbr) if False else None
    # _compute_similarity only uses the static _gather_neighbors; call it unbound:
    class _H:
        _gather_neighbors = staticmethod(M._gather_neighbors)
    s = M._compute_similarity(_H, y, nbr)
    out['s'] = s.squeeze(-1).numpy()
    for steps in (1, 3, 5):
        x = z
        I = torch.eye(H, dtype=torch.double)
        C = c.t() @ c
        for _ in range(steps):                      # continuous_crf_conv_big.py:68-72 verbatim ops
            xx = M._gather_neighbors(x, nbr)
            xx = (s * xx).sum(dim=2)
            xx = z + xx.matmul(C)
            x = xx.matmul((I + C).inverse())
        out['x_T%d' % steps] = x.numpy()
    save('g2_meanfield_fp64.npz', **out)


# ----------------------------------------------------------------------------- G3/G4: PointConv, ResNetBBlock
def g3_pointconv(models):
    from models.point_conv_big import PointConv
    B, N, K, d = 2, 256, 16, 8
    pos = np.stack([S.make_cloud(300 + b, N) for b in range(B)])
    nbr = ref_knn(pos, pos, K)
    choice = S.permutation(3, 'g3choice', N)[: N // 4]
    sub_pos = np.ascontiguousarray(pos[:, choice])
    sub_idx = np.ascontiguousarray(nbr[:, choice])
    x = torch.from_numpy(S.uniform(3, 'g3x', (B, N, d)))
    out = dict(pos=pos, neighbor_idx=nbr.astype(np.int16), sub_pos=sub_pos,
               sub_idx=sub_idx.astype(np.int16), x=x.numpy())
    for form in ('plain', 'strided'):
        M = N if form == 'plain' else N // 4
        gout = torch.from_numpy(S.uniform(3, 'g3g' + form, (B, M, d)))
        out[form + '/gout'] = gout.numpy()
        for mode in ('train', 'eval'):
            m = PointConv(d)
            sd = load_seeded(m, 11)
            m.train(mode == 'train')
            xi = x.clone().requires_grad_(True)
            if form == 'plain':
                y = m(xi, torch.from_numpy(pos), torch.from_numpy(nbr))
            else:
                y = m(xi, (torch.from_numpy(pos), torch.from_numpy(sub_pos)), torch.from_numpy(sub_idx))
            (y * gout).sum().backward()
            tag = '%s_%s' % (form, mode)
            out[tag + '/out'] = y.detach().numpy()
            out[tag + '/d_x'] = xi.grad.numpy()
            out.update(pack(tag + '/grad', grads_of(m)))
            if mode == 'train':
                out.update(pack(tag + '/buf', buffers_after(m)))
    out.update(pack('sd', sd))
    save('g3_pointconv.npz', **out)


def g4_resblock(models):
    from models.point_conv_big import ResNetBBlock
    B, N, K = 2, 256, 16
    pos = np.stack([S.make_cloud(400 + b, N) for b in range(B)])
    nbr = ref_knn(pos, pos, K)
    choice = S.permutation(4, 'g4choice', N)[: N // 4]
    sub_pos = np.ascontiguousarray(pos[:, choice])
    sub_idx = np.ascontiguousarray(nbr[:, choice])
    out = dict(pos=pos, neighbor_idx=nbr.astype(np.int16), sub_pos=sub_pos,
               sub_idx=sub_idx.astype(np.int16))
    for name, cin, cout, strided in (('a', 6, 32, False), ('b', 32, 32, False), ('c', 32, 64, True)):
        M = N // 4 if strided else N
        x = torch.from_numpy(S.uniform(4, 'g4x' + name, (B, N, cin)))
        gout = torch.from_numpy(S.uniform(4, 'g4g' + name, (B, M, cout)))
        out[name + '/x'] = x.numpy()
        out[name + '/gout'] = gout.numpy()
        for mode in (('train', 'eval') if name != 'b' else ('train',)):
            m = ResNetBBlock(cin, cout)
            sd = load_seeded(m, 13)
            m.train(mode == 'train')
            xi = x.clone().requires_grad_(True)
            if strided:
                y = m(xi, (torch.from_numpy(pos), torch.from_numpy(sub_pos)), torch.from_numpy(sub_idx))
            else:
                y = m(xi, torch.from_numpy(pos), torch.from_numpy(nbr))
            (y * gout).sum().backward()
            tag = '%s_%s' % (name, mode)
            out[tag + '/out'] = y.detach().numpy()
            out[tag + '/d_x'] = xi.grad.numpy()
            out.update(pack(tag + '/grad', grads_of(m)))
        out.update(pack(name + '/sd', sd))
    save('g4_resblock.npz', **out)


# ----------------------------------------------------------------------------- G5: whole net + training-step contract
def g5_pointconvbig(models):
    B, N, Cin, ncls, steps = 2, 4096, 6, 13, 3
    pos = np.stack([S.make_cloud(500 + b, N, box=(2.0, 2.0, 1.0)) for b in range(B)])
    ms_np = S.build_multiscale(pos, ref_knn, seed=5)
    feats = np.concatenate([pos, S.uniform(5, 'g5rgb', (B, N, 3), 0.0, 1.0)], -1).astype(np.float32)
    labels = S.integers(5, 'g5y', (B, N), 0, ncls + 1)          # 0 = unlabeled -> ignore_index -1
    cw = S.uniform(5, 'g5cw', (ncls,), 0.5, 2.0)

    ms = [Obj(**{k: torch.from_numpy(v) for k, v in lvl.items()}) for lvl in ms_np]
    data = Obj(x=torch.from_numpy(feats), multiscale=ms)
    out = dict(pos=pos, feats=feats, labels=labels.astype(np.int16), class_weights=cw,
               steps=np.int64(steps))
    for i, lvl in enumerate(ms_np):
        for k in ('neighbor_idx', 'sub_idx', 'up_idx'):
            out['ms%d/%s' % (i, k)] = lvl[k].astype(np.int16)
    rows = np.sort(S.permutation(5, 'g5rows', B * N)[:768])
    out['rows'] = rows.astype(np.int32)
    for use_crf in (True, False):
        net = models.PointConvBig(Cin, ncls, use_crf=use_crf, steps=steps)
        shapes = shapes_of(net)
        load_seeded(net, 17)
        tagc = 'crf' if use_crf else 'ups'
        out[tagc + '/keys'] = np.array(sorted(shapes))
        out[tagc + '/shapes'] = np.array([','.join(map(str, shapes[k])) for k in sorted(shapes)])
        # eval
        net.eval()
        with torch.no_grad():
            lg = net(data).numpy()
        out[tagc + '_eval/logits_rows'] = lg[rows]
        out[tagc + '_eval/logits_colsum'] = lg.astype(np.float64).sum(0)
        # train step (trainval.py:99-105), dropout mask captured from the module
        net.train()
        mask = {}

        def hook(mod, inp, outp):
            mask['m'] = (outp != 0).to(torch.uint8)
        h = net.classifier[1].register_forward_hook(hook)
        torch.manual_seed(1234)
        logits = net(data)
        h.remove()
        y = torch.from_numpy(labels.reshape(-1)).long() - 1
        loss = torch.nn.functional.cross_entropy(logits, y, weight=torch.from_numpy(cw), ignore_index=-1)
        loss.backward()
        out[tagc + '_train/logits_rows'] = logits.detach().numpy()[rows]
        out[tagc + '_train/logits_colsum'] = logits.detach().numpy().astype(np.float64).sum(0)
        out[tagc + '_train/loss'] = loss.detach().numpy()
        out[tagc + '_train/dropout_mask'] = np.packbits(mask['m'].numpy().reshape(-1))
        g = grads_of(net)
        for k, v in g.items():
            v = v.numpy().reshape(-1).astype(np.float64)
            out['%s_train/gnorm/%s' % (tagc, k)] = np.sqrt((v * v).sum())
            out['%s_train/gproj/%s' % (tagc, k)] = S.projections(17, k, v.size).astype(np.float64) @ v
            if v.size <= 1024:
                out['%s_train/grad/%s' % (tagc, k)] = g[k].numpy()
    save('g5_pointconvbig.npz', **out)


# ----------------------------------------------------------------------------- G6: kNN, G7: grid subsampling
def g6_knn(models):
    out = {}
    pts = np.stack([S.make_cloud(600 + b, 2048) for b in range(2)])
    qry = np.stack([S.make_cloud(610 + b, 512) for b in range(2)])
    out.update(pts=pts, qry=qry)
    for K in (1, 16, 32):
        a = onative.ref_knn_batch(pts, pts, K)
        b = onative.ref_knn_batch(pts, pts, K, omp=True)
        assert np.array_equal(a, b)
        assert np.array_equal(onative.ref_knn(pts[0], pts[0], K), a[0])
        assert np.array_equal(onative.ref_knn(pts[0], pts[0], K, omp=True), a[0])
        out['self_K%d' % K] = a.astype(np.int16)
        out['cross_K%d' % K] = onative.ref_knn_batch(pts, qry, K).astype(np.int16)
    # tie-heavy lattice: only the sorted distance rows are well defined
    lat = (S.integers(6, 'lat', (1024, 3), 0, 16) / 16.0).astype(np.float32)
    idx = onative.ref_knn(lat, lat, 16)
    out['lattice_pts'] = lat
    out['lattice_dists'] = onative.knn_sq_dists(lat, lat, idx)
    # S3DIS-like cloud: one point per 4 cm voxel, jittered (BASELINE config-2 recipe, small)
    save('g6_knn.npz', **out)


def g7_grid(models):
    out = {}
    N = 6000
    pts = (S.make_cloud(700, N, box=(3.0, 2.0, 1.5)) - np.float32(0.7)).astype(np.float32)
    feats = S.uniform(7, 'f', (N, 3), 0.0, 255.0)
    lab1 = S.integers(7, 'l1', (N,), 0, 13).astype(np.int32)
    lab2 = S.integers(7, 'l2', (N, 2), 0, 8).astype(np.int32)
    out.update(pts=pts, feats=feats, lab1=lab1, lab2=lab2)
    keys = onative.grid_keys(pts, 0.1)          # per input point (oracle arithmetic == reference)
    for name, f, c, dl in (('all', feats, lab1, 0.1), ('two', feats, lab2, 0.1), ('ponly', None, None, 0.1),
                           ('fonly', feats, None, 0.06), ('conly', None, lab1, 0.25)):
        rp, rf, rc = onative.ref_grid_subsample(pts, f, c, dl)
        out[name + '/dl'] = np.float32(dl)
        out[name + '/pts'] = rp                    # reference row order kept as produced
        if rf is not None:
            out[name + '/feats'] = rf
        if rc is not None:
            out[name + '/classes'] = rc
    save('g7_grid.npz', **out)


# ----------------------------------------------------------------------------- G8: sparse (edge-list) operators
def g8_sparse(models):
    from models.continuous_crf_conv import ContinuousGaussianCRFConv as SparseCRF, GuideGaussianCRFConv
    from models.point_conv import DepthwiseSeparablePointConv
    N, K, NA = 320, 12, 180
    pos = np.concatenate([S.make_cloud(800, NA), S.make_cloud(801, N - NA) + np.float32(3.0)])
    batch = np.concatenate([np.zeros(NA, np.int64), np.ones(N - NA, np.int64)])
    # kNN graph without self loops inside each cloud: target i receives from its K nearest j
    nb0 = onative.ref_knn(pos[:NA], pos[:NA], K + 1)[:, 1:]
    nb1 = onative.ref_knn(pos[NA:], pos[NA:], K + 1)[:, 1:] + NA
    nbr = np.concatenate([nb0, nb1])
    tgt = np.repeat(np.arange(N), K)
    src = nbr.reshape(-1)
    # ragged variant: drop a seeded 30 % of the edges
    keep = S.uniform(8, 'keep', (tgt.size,), 0, 1) > 0.3
    out = dict(pos=pos, batch=batch, tgt=tgt.astype(np.int16), src=src.astype(np.int16), keep=keep)
    x = torch.from_numpy(S.uniform(8, 'x', (N, 32)))
    y = torch.from_numpy(S.uniform(8, 'y', (N, 16)))
    out.update(x=x.numpy(), y=y.numpy())
    for gname, t, s in (('full', tgt, src), ('ragged', tgt[keep], src[keep])):
        tt, ss = torch.from_numpy(t), torch.from_numpy(s)
        for steps, mode in (((3, 'train'),) if gname == 'full' else ((3, 'train'), (1, 'eval'))):
            if True:
                tag = '%s_T%d_%s' % (gname, steps, mode)
                # explicit-edge CRFConv aggregates at edge_index[0]
                m = SparseCRF(32, 16, None, 16, steps=steps)
                sd = load_seeded(m, 19)
                m.train(mode == 'train')
                xi, yi = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
                o = m(xi, yi, torch.from_numpy(pos), torch.stack([tt, ss]))
                gout = torch.from_numpy(S.uniform(8, 'g' + tag, tuple(o.shape)))
                (o * gout).sum().backward()
                out['crf/' + tag + '/out'] = o.detach().numpy()
                out['crf/' + tag + '/gout'] = gout.numpy()
                out['crf/' + tag + '/d_x'] = xi.grad.numpy()
                out['crf/' + tag + '/d_y'] = yi.grad.numpy()
                out.update(pack('crf/' + tag + '/grad', grads_of(m)))
                # guided CRFConv: graph from (stubbed) radius_graph -> (col=src, row=tgt)
                g = GuideGaussianCRFConv(32, 16, 8, radius=0.1, kernel_size=K, steps=steps)
                sdg = load_seeded(g, 23)
                g.train(mode == 'train')
                _INJECTED_GRAPH['edge_index'] = torch.stack([ss, tt])
                xi, yi = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
                o = g(xi, yi, torch.from_numpy(pos), torch.from_numpy(batch))
                gout = torch.from_numpy(S.uniform(8, 'gg' + tag, tuple(o.shape)))
                (o * gout).sum().backward()
                out['guide/' + tag + '/out'] = o.detach().numpy()
                out['guide/' + tag + '/gout'] = gout.numpy()
                out['guide/' + tag + '/d_x'] = xi.grad.numpy()
                out['guide/' + tag + '/d_y'] = yi.grad.numpy()
                out.update(pack('guide/' + tag + '/grad', grads_of(g)))
    out.update(pack('crf/sd', sd))
    out.update(pack('guide/sd', sdg))
    # depth-wise separable point conv: symmetric (self loops re-added) and bipartite
    xs = torch.from_numpy(S.uniform(8, 'xs', (N, 16)))
    out['dsconv/x'] = xs.numpy()
    choice = np.sort(S.permutation(8, 'dschoice', N)[:80])
    out['dsconv/choice'] = choice.astype(np.int16)
    tt, ss = torch.from_numpy(tgt), torch.from_numpy(src)
    # bipartite: each coarse point (row index into choice) gathers its fine-level neighbours
    bi_row = np.repeat(np.arange(80), K)
    bi_col = nbr[choice].reshape(-1)
    for form, cin, cout in (('sym_same', 16, 16), ('sym_proj', 16, 32), ('bip', 16, 32)):
        for mode in ('train', 'eval'):
            m = DepthwiseSeparablePointConv(cin, cout)
            sdd = load_seeded(m, 29)
            m.train(mode == 'train')
            xi = xs.clone().requires_grad_(True)
            if form == 'bip':
                ei = torch.stack([torch.from_numpy(bi_col), torch.from_numpy(bi_row)])
                o = m(xi, (torch.from_numpy(pos), torch.from_numpy(pos[choice])), ei)
            else:
                o = m(xi, torch.from_numpy(pos), torch.stack([ss, tt]))
            gout = torch.from_numpy(S.uniform(8, 'gd' + form + mode, tuple(o.shape)))
            (o * gout).sum().backward()
            tag = 'dsconv/%s_%s' % (form, mode)
            out[tag + '/out'] = o.detach().numpy()
            out[tag + '/gout'] = gout.numpy()
            out[tag + '/d_x'] = xi.grad.numpy()
            out.update(pack(tag + '/grad', grads_of(m)))
        out.update(pack('dsconv/%s/sd' % form, sdd))
    save('g8_sparse.npz', **out)


def g11_shapenet(models):
    """utils/metrics.py runningScoreShapeNet (pure numpy) on seeded part labels of eight shapes."""
    metrics = import_reference_file('utils/metrics.py', 'ref_metrics')
    rs = metrics.runningScoreShapeNet()
    out, cats, ious = {}, [0, 4, 4, 10, 15, 8, 0, 13], []
    for i, c in enumerate(cats):
        name = [k for k, v in rs.obj_classes.items() if v == c][0]
        parts = np.array(rs.seg_classes[name])
        yt = parts[S.integers(11, 'yt%d' % i, (2048,), 0, len(parts))]
        yp = parts[S.integers(11, 'yp%d' % i, (2048,), 0, len(parts))]
        yp = np.where(S.uniform(11, 'k%d' % i, (2048,), 0, 1) < 0.6, yt, yp)
        if i == 3:
            yp[:] = parts[0]                                   # parts that are never predicted
        out['yt%d' % i] = yt.astype(np.int16)
        out['yp%d' % i] = yp.astype(np.int16)
        ious.append(rs.update(yt, yp, c))
    p, mp, cls = rs.get_scores()
    out.update(cats=np.array(cats), ious=np.array(ious), pIoU=np.float64(p), mpIoU=np.float64(mp),
               cls=np.array([cls[k] for k in sorted(cls)]), cls_names=np.array(sorted(cls)))
    save('g11_shapenet_score.npz', **out)


def g10_discrete(models):
    """models/discrete_crf_conv.py run as it is; the radius graph is injected (torch_cluster absent), scatter_add is
    the index_add restatement above."""
    from models.discrete_crf_conv import DiscreteCRFConv
    N, K, NA, L, D = 300, 10, 170, 13, 6
    pos = np.concatenate([S.make_cloud(1000, NA), S.make_cloud(1001, N - NA) + np.float32(3.0)])
    nb0 = onative.ref_knn(pos[:NA], pos[:NA], K + 1)[:, 1:]
    nb1 = onative.ref_knn(pos[NA:], pos[NA:], K + 1)[:, 1:] + NA
    tgt = np.repeat(np.arange(N), K)
    src = np.concatenate([nb0, nb1]).reshape(-1)
    keep = S.uniform(10, 'keep', (tgt.size,), 0, 1) > 0.25         # ragged degrees, some isolated targets possible
    tgt, src = tgt[keep], src[keep]
    out = dict(pos=pos, tgt=tgt.astype(np.int16), src=src.astype(np.int16))
    logit = torch.from_numpy(S.uniform(10, 'logit', (N, L), -2, 2))
    f = torch.from_numpy(S.uniform(10, 'f', (N, D), 0, 1))
    out.update(logit=logit.numpy(), f=f.numpy())
    for steps, hidden, kernels in ((1, 64, 5), (3, 64, 5), (5, 16, 3), (2, 96, 2)):
        tag = 'T%d_H%d_G%d' % (steps, hidden, kernels)
        m = DiscreteCRFConv(L, D, hidden_channels=hidden, num_kernels=kernels, radius=0.2, kernel_size=K, steps=steps)
        sd = {'F': torch.from_numpy(S.uniform(10, 'F' + tag, (kernels, D, hidden), 0, 0.35)),
              'W': torch.from_numpy(S.uniform(10, 'W' + tag, (kernels, 1), 0.05, 0.6)),
              'C': torch.from_numpy((np.eye(L) + 0.3 * S.uniform(10, 'C' + tag, (L, L))).astype(np.float32))}
        m.load_state_dict(sd, strict=True)
        _INJECTED_GRAPH['edge_index'] = torch.stack([torch.from_numpy(src), torch.from_numpy(tgt)])
        li, fi = logit.clone().requires_grad_(True), f.clone().requires_grad_(True)
        q = m(torch.from_numpy(pos), torch.softmax(li, dim=-1), f=fi, batch=None)
        gout = torch.from_numpy(S.uniform(10, 'g' + tag, tuple(q.shape)))
        (torch.log(q) * gout).sum().backward()
        out.update(pack(tag + '/sd', sd))
        out.update(pack(tag + '/grad', grads_of(m)))
        out[tag + '/q'] = q.detach().numpy()
        out[tag + '/gout'] = gout.numpy()
        out[tag + '/d_logit'] = li.grad.numpy()
        out[tag + '/d_f'] = fi.grad.numpy()
    save('g10_discrete.npz', **out)


def import_reference_file(relpath, name, stubs=()):
    """Loads ONE reference source file as a module (its package __init__ would pull in half of torch_geometric)."""
    import importlib.util
    for mod_name, attrs in stubs:
        m = types.ModuleType(mod_name)
        m.__dict__.update(attrs)
        sys.modules[mod_name] = m
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def g9_eval(models):
    """utils/metrics.py (pure numpy) and Semantic3D._get_random (datasets/semantic3d_dataset.py:423-460) run as they
    are; the dataset module's module-scope imports that the method never touches are empty stubs, `Data` is an
    attribute bag (torch_geometric.data.Data is only used as one at :456)."""
    from sklearn.neighbors import KDTree
    metrics = import_reference_file('utils/metrics.py', 'ref_metrics')
    out = {}
    n_cls = 13
    yt = S.integers(9, 'yt', (3, 5000), -1, n_cls + 1)          # includes ignore (-1) and an out-of-range label
    yp = S.integers(9, 'yp', (3, 5000), 0, n_cls)
    rs = metrics.runningScore(n_cls, ignore_index=-1)
    rs.update(yt, yp)
    rs.update(yt[0], yp[1])
    sc, cls_iu = rs.get_scores()
    out.update(m_yt=yt, m_yp=yp, m_hist=rs.confusion_matrix, m_scores=np.array([sc[k] for k in sorted(sc)]),
               m_score_names=np.array(sorted(sc)), m_cls_iu=np.array([cls_iu[c] for c in range(n_cls)]))
    rs2 = metrics.runningScore(n_cls, ignore_index=3)
    rs2.update(yt.reshape(-1), yp.reshape(-1))
    out['m_hist_ignore3'] = rs2.confusion_matrix

    class Bag:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    empty = lambda *names: {n: None for n in names}
    ds = import_reference_file('datasets/semantic3d_dataset.py', 'ref_semantic3d', stubs=[
        ('plyfile', empty('PlyData')),
        ('torch_geometric.data', dict(Data=Bag, Dataset=object, InMemoryDataset=object)),
        ('torch_points_kernels', {}), ('torch_points_kernels.points_cpu', {}), ('torch_points_kernels.points_cuda', {}),
        ('torch_points3d.datasets', {}), ('torch_points3d.datasets.batch', empty('SimpleBatch')),
        ('torch_points3d.datasets.multiscale_data', empty('MultiScaleData', 'MultiScaleBatch')),
        ('utils', empty('cpp_subsampling', 'nearest_neighbors', 'read_ply', 'write_ply', 'Plot')),
    ])
    # two clouds, crops of 1500 points, 6 consecutive draws (cloud choice, possibility bookkeeping and all)
    clouds = [S.make_cloud(900, 6000, box=(6.0, 5.0, 2.0)), S.make_cloud(901, 4000, box=(4.0, 4.0, 2.0))]
    labels = [S.integers(9, 'lab0', (6000,), 1, 9), S.integers(9, 'lab1', (4000,), 1, 9)]
    rgb = [S.uniform(9, 'rgb0', (6000, 3), 0, 1), S.uniform(9, 'rgb1', (4000, 3), 0, 1)]
    cw = 1.0 / (S.uniform(9, 'cw', (1, 8), 0.02, 0.5).astype(np.float64) + 0.02)
    poss0 = [S.uniform(9, 'p0', (6000,), -1, 1).astype(np.float64) * 1e-3, S.uniform(9, 'p1', (4000,), -1, 1).astype(np.float64) * 1e-3]
    for split in ('train', 'test'):
        fake = Bag(min_possibility=[float(p.min()) for p in poss0], possibility=[p.copy() for p in poss0],
                   input_trees=[KDTree(c, leaf_size=50) for c in clouds], input_rgb=rgb, input_labels=labels,
                   label_to_idx={l: i for i, l in enumerate(range(1, 9))}, class_weight=cw, num_points=1500, split=split)
        np.random.seed(1234)
        for draw in range(6):
            state = np.random.get_state()
            d = ds.Semantic3D._get_random(fake)
            np.random.set_state(state)
            noise = np.random.normal(scale=3.5 / 10, size=(1, 3))      # the draw _get_random made first (:429)
            np.random.shuffle(np.arange(1500))                           # ... and its shuffle (:434), to stay in step
            tag = 's_%s_%d_' % (split, draw)
            out[tag + 'noise'] = noise.reshape(-1)
            out[tag + 'cloud'] = d.cloud_idx.numpy()
            out[tag + 'point_idx'] = d.point_idx.numpy().astype(np.int32)
            out[tag + 'pos'] = d.pos.numpy()
            out[tag + 'y'] = d.y.numpy().astype(np.int16)
            out[tag + 'rgb'] = d.rgb.numpy()
            out[tag + 'min_possibility'] = np.array(fake.min_possibility)
        out['s_%s_possibility0' % split] = fake.possibility[0]
        out['s_%s_possibility1' % split] = fake.possibility[1]
    out.update(s_cloud0=clouds[0], s_cloud1=clouds[1], s_labels0=labels[0].astype(np.int16), s_labels1=labels[1].astype(np.int16),
               s_rgb0=rgb[0], s_rgb1=rgb[1], s_cw=cw, s_poss0=poss0[0], s_poss1=poss0[1])
    save('g9_eval.npz', **out)


if __name__ == '__main__':
    torch.set_num_threads(4)
    torch.manual_seed(0)
    onative.build(ref=True)
    models = import_reference_models()
    only = set(sys.argv[1:])
    for fn in (g1_crfconv, g2_meanfield_fp64, g3_pointconv, g4_resblock, g5_pointconvbig, g6_knn,
               g7_grid, g8_sparse, g9_eval, g10_discrete, g11_shapenet):
        if not only or fn.__name__.split('_')[0] in only:
            fn(models)
