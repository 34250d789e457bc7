"""CPU: the oracle (oracle/) against the fixtures captured from the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle; the HIP path is then compared with
the oracle in the -m gpu tests."""
import numpy as np
import pytest
import torch

import _seeded as S
from conftest import sub
from oracle import crf_oracle as O
from oracle import native as onative

torch.set_num_threads(4)
OUT_TOL = 2e-6      # outputs (fp32, same op sequence, possibly different reduction order)
GRAD_TOL = 2e-5


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x.to(dtype) if dtype is not None else x


def params(sd_np, requires_grad=True):
    sd = {}
    for k, v in sd_np.items():
        x = t(v).clone()
        if requires_grad and x.is_floating_point() and 'running_' not in k:
            x.requires_grad_(True)
        sd[k] = x
    return sd


def close(a, b, tol, what=''):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    err = np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert err <= tol, '%s: rel-max err %.3e > %.1e' % (what, err, tol)


def check_grads(sd, gold, prefix, tol=GRAD_TOL):
    g = sub(gold, prefix)
    assert g, prefix
    for k, v in g.items():
        close(sd[k].grad, v, tol, prefix + '/' + k)


@pytest.mark.parametrize('steps,mode', [(1, 'train'), (3, 'train'), (3, 'eval'), (5, 'eval')])
def test_crfconv_golden(golden, steps, mode):
    g = golden('g1_crfconv.npz')
    sd = params(sub(g, 'sd'))
    u = t(g['unary']).requires_grad_(True)
    p = t(g['pairwise']).requires_grad_(True)
    out = O.crf_conv(sd, '', u, p, t(g['up_idx'], torch.long), t(g['neighbor_idx'], torch.long), steps,
                     mode == 'train')
    tag = 'T%d_%s' % (steps, mode)
    close(out, g[tag + '/out'], OUT_TOL, 'out')
    (out * t(g['gout'])).sum().backward()
    close(u.grad, g[tag + '/d_unary'], GRAD_TOL, 'd_unary')
    close(p.grad, g[tag + '/d_pairwise'], GRAD_TOL, 'd_pairwise')
    check_grads(sd, g, tag + '/grad')
    if mode == 'train':
        for k, v in sub(g, tag + '/buf').items():
            if 'num_batches' not in k:
                close(sd[k], v, 1e-6, k)


def test_meanfield_fp64_golden(golden):
    g = golden('g2_meanfield_fp64.npz')
    z, y, c = t(g['z']), t(g['y']), t(g['c'])
    nbr = t(g['nbr'], torch.long)
    close(O.crf_similarity(y, nbr), g['s'], 1e-13, 's')
    for steps in (1, 3, 5):
        close(O.crf_meanfield(z, y, nbr, c, steps), g['x_T%d' % steps], 1e-12, 'x_T%d' % steps)
    # the fp32 evaluation of the same thing stays within 1e-5 of the fp64 anchor
    x32 = O.crf_meanfield(z.float(), y.float(), nbr, c.float(), 5)
    close(x32.double(), g['x_T5'], 1e-5, 'fp32 vs fp64')


@pytest.mark.parametrize('form', ['plain', 'strided'])
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_pointconv_golden(golden, form, mode):
    g = golden('g3_pointconv.npz')
    sd = params(sub(g, 'sd'))
    x = t(g['x']).requires_grad_(True)
    if form == 'plain':
        out = O.point_conv(sd, '', x, t(g['pos']), t(g['neighbor_idx'], torch.long), mode == 'train')
    else:
        out = O.point_conv(sd, '', x, (t(g['pos']), t(g['sub_pos'])), t(g['sub_idx'], torch.long),
                           mode == 'train')
    tag = '%s_%s' % (form, mode)
    close(out, g[tag + '/out'], OUT_TOL, 'out')
    (out * t(g[form + '/gout'])).sum().backward()
    close(x.grad, g[tag + '/d_x'], GRAD_TOL, 'd_x')
    check_grads(sd, g, tag + '/grad')


@pytest.mark.parametrize('name,mode', [('a', 'train'), ('a', 'eval'), ('b', 'train'), ('c', 'train'), ('c', 'eval')])
def test_resblock_golden(golden, name, mode):
    g = golden('g4_resblock.npz')
    sd = params(sub(g, name + '/sd'))
    x = t(g[name + '/x']).requires_grad_(True)
    if name == 'c':
        out = O.resnet_block(sd, '', x, (t(g['pos']), t(g['sub_pos'])), t(g['sub_idx'], torch.long), mode == 'train')
    else:
        out = O.resnet_block(sd, '', x, t(g['pos']), t(g['neighbor_idx'], torch.long), mode == 'train')
    tag = '%s_%s' % (name, mode)
    close(out, g[tag + '/out'], OUT_TOL, 'out')
    (out * t(g[name + '/gout'])).sum().backward()
    close(x.grad, g[tag + '/d_x'], GRAD_TOL, 'd_x')
    check_grads(sd, g, tag + '/grad')


def g5_inputs(g):
    B, N = g['pos'].shape[:2]
    ms = []
    pos = g['pos']
    for i in range(5):
        lvl = {'pos': t(pos)}
        for k in ('neighbor_idx', 'sub_idx', 'up_idx'):
            lvl[k] = t(g['ms%d/%s' % (i, k)], torch.long)
        ms.append(lvl)
        n = pos.shape[1]
        choice = S.permutation(5, 'choice%d' % i, n)[: n // (4, 4, 4, 4, 2)[i]]
        pos = np.ascontiguousarray(pos[:, choice])
    return ms


def g5_state(g, tagc, seed=17):
    shapes = {k: tuple(int(s) for s in sh.split(',')) if sh else () for k, sh in zip(g[tagc + '/keys'], g[tagc + '/shapes'])}
    return S.fill_state_dict(shapes, seed)


@pytest.mark.parametrize('use_crf', [True, False])
def test_pointconvbig_golden(golden, use_crf):
    """Whole network + the training-step contract (trainval.py:99-105) at B=2, N=4096."""
    g = golden('g5_pointconvbig.npz')
    tagc = 'crf' if use_crf else 'ups'
    ms = g5_inputs(g)
    sd = params(g5_state(g, tagc))
    sd = {k: v for k, v in sd.items()}
    x = t(g['feats'])
    rows = g['rows']
    steps = int(g['steps'])
    with torch.no_grad():
        sd_eval = {k: v.detach().clone() for k, v in sd.items()}
        lg = O.pointconv_resnet(sd_eval, x, ms, steps, False, use_crf).numpy()
    close(lg[rows], g[tagc + '_eval/logits_rows'], 5e-6, 'eval logits')
    close(lg.astype(np.float64).sum(0), g[tagc + '_eval/logits_colsum'], 5e-6, 'eval colsum')

    B, N = g['pos'].shape[:2]
    mask = np.unpackbits(g[tagc + '_train/dropout_mask'])[: B * N * 128].reshape(B, N, 128)
    logits = O.pointconv_resnet(sd, x, ms, steps, True, use_crf, dropout_mask=t(mask).float())
    loss = O.training_loss(logits, t(g['labels'], torch.long), t(g['class_weights']))
    loss.backward()
    close(logits.detach().numpy()[rows], g[tagc + '_train/logits_rows'], 1e-5, 'train logits')
    close(loss, g[tagc + '_train/loss'], 1e-6, 'loss')
    for k, v in sub(g, tagc + '_train/gnorm').items():
        gr = sd[k].grad.numpy().reshape(-1).astype(np.float64)
        nrm = float(v)
        assert abs(np.sqrt((gr * gr).sum()) - nrm) <= 2e-4 * max(nrm, 1e-3), k
        proj = S.projections(17, k, gr.size).astype(np.float64) @ gr
        assert np.abs(proj - g['%s_train/gproj/%s' % (tagc, k)]).max() <= 2e-4 * max(nrm, 1e-3) * np.sqrt(gr.size), k
    for k, v in sub(g, tagc + '_train/grad').items():
        close(sd[k].grad, v, 5e-4, k)


# ------------------------------------------------------------------ native checkers
@pytest.mark.parametrize('K', [1, 16, 32])
def test_knn_oracle_golden(golden, K):
    g = golden('g6_knn.npz')
    a = onative.oracle_knn_batch(g['pts'], g['pts'], K)
    b = onative.oracle_knn_batch(g['pts'], g['qry'], K)
    assert np.array_equal(a, g['self_K%d' % K].astype(np.int64))
    assert np.array_equal(b, g['cross_K%d' % K].astype(np.int64))
    assert np.array_equal(onative.oracle_knn(g['pts'][1], g['qry'][1], K), b[1])


def test_knn_oracle_lattice_distances(golden):
    g = golden('g6_knn.npz')
    lat = g['lattice_pts']
    idx = onative.oracle_knn(lat, lat, 16)
    assert np.array_equal(onative.knn_sq_dists(lat, lat, idx), g['lattice_dists'])


def rekey(pts_in, dl, rows_pts):
    """voxel key of each OUTPUT row: the key of the input voxel whose barycentre it is."""
    keys = onative.grid_keys(pts_in, dl)
    op, _, _, okeys = onative.oracle_grid_subsample(pts_in, None, None, dl)
    # barycentres are unique per voxel; match rows through exact float triples
    lut = {tuple(p): k for p, k in zip(map(tuple, op), okeys)}
    return np.array([lut[tuple(p)] for p in map(tuple, rows_pts)], dtype=np.uint64)


@pytest.mark.parametrize('name', ['all', 'two', 'ponly', 'fonly', 'conly'])
def test_grid_oracle_golden(golden, name):
    g = golden('g7_grid.npz')
    dl = float(g[name + '/dl'])
    f = g['feats'] if name in ('all', 'two', 'fonly') else None
    c = {'all': g['lab1'], 'two': g['lab2'], 'conly': g['lab1']}.get(name)
    op, of, oc, okeys = onative.oracle_grid_subsample(g['pts'], f, c, dl)
    rp = g[name + '/pts']
    assert op.shape == rp.shape
    order = np.argsort(rekey(g['pts'], dl, rp), kind='stable')   # reference rows -> ascending key
    assert np.array_equal(op, rp[order])                          # barycentres bit-exact
    if f is not None:
        assert np.array_equal(of, g[name + '/feats'][order])      # arrival-order float sums bit-exact
    if c is not None:
        rc = g[name + '/classes'][order]
        # labels must agree wherever the vote is not tied (ties: reference = hash-map order)
        cc = c if c.ndim == 2 else c[:, None]
        keys = onative.grid_keys(g['pts'], dl)
        tied = np.zeros(rc.shape, dtype=bool)
        pos_of = {k: i for i, k in enumerate(okeys)}
        votes = [[{} for _ in range(cc.shape[1])] for _ in range(len(okeys))]
        for i, k in enumerate(keys):
            for l in range(cc.shape[1]):
                d = votes[pos_of[k]][l]
                d[cc[i, l]] = d.get(cc[i, l], 0) + 1
        for r in range(len(okeys)):
            for l in range(cc.shape[1]):
                v = sorted(votes[r][l].values())
                tied[r, l] = len(v) > 1 and v[-1] == v[-2]
        assert np.array_equal(oc[~tied], rc[~tied])
        assert (~tied).mean() > 0.3


def test_ref_libs_agree_with_oracle_when_present():
    if not onative.have_ref():
        pytest.skip('oracle/_ref not built (needs /root/reference)')
    rng = np.random.default_rng(3)
    pts = rng.random((2, 1500, 3), dtype=np.float32)
    assert np.array_equal(onative.oracle_knn_batch(pts, pts, 16), onative.ref_knn_batch(pts, pts, 16, omp=True))


# ------------------------------------------------------------------ sparse (edge-list) twins
def _g8_graph(g, gname):
    tgt = g['tgt'].astype(np.int64)
    src = g['src'].astype(np.int64)
    if gname == 'ragged':
        tgt, src = tgt[g['keep']], src[g['keep']]
    return torch.from_numpy(tgt), torch.from_numpy(src)


@pytest.mark.parametrize('gname,steps,mode', [('full', 3, 'train'), ('ragged', 3, 'train'), ('ragged', 1, 'eval')])
def test_sparse_crf_oracle_golden(golden, gname, steps, mode):
    """oracle's edge-list mean field vs the reference's sparse modules run through third-party stubs
    ("parity unpinned" at the torch_geometric / torch_scatter boundary)."""
    g = golden('g8_sparse.npz')
    tgt, src = _g8_graph(g, gname)
    tag = '%s_T%d_%s' % (gname, steps, mode)
    for kind in ('crf', 'guide'):
        sd = params(sub(g, kind + '/sd'))
        x = t(g['x']).requires_grad_(True)
        y = t(g['y']).requires_grad_(True)
        if kind == 'crf':
            out = O.sparse_crf_conv(sd, '', x, y, torch.stack([tgt, src]), steps, mode == 'train')
        else:
            out = O.guide_crf_conv(sd, '', x, y, tgt, src, steps, mode == 'train')
        close(out, g['%s/%s/out' % (kind, tag)], OUT_TOL, kind + ' out')
        (out * t(g['%s/%s/gout' % (kind, tag)])).sum().backward()
        close(x.grad, g['%s/%s/d_x' % (kind, tag)], GRAD_TOL, kind + ' d_x')
        close(y.grad, g['%s/%s/d_y' % (kind, tag)], GRAD_TOL, kind + ' d_y')
        check_grads(sd, g, '%s/%s/grad' % (kind, tag))


@pytest.mark.parametrize('form', ['sym_same', 'sym_proj', 'bip'])
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_ds_point_conv_oracle_golden(golden, form, mode):
    g = golden('g8_sparse.npz')
    sd = params(sub(g, 'dsconv/%s/sd' % form))
    x = t(g['dsconv/x']).requires_grad_(True)
    pos = t(g['pos'])
    tgt, src = _g8_graph(g, 'full')
    if form == 'bip':
        choice = g['dsconv/choice'].astype(np.int64)
        K = len(tgt) // len(pos)
        nbr = src.reshape(len(pos), K)
        ei = torch.stack([nbr[choice].reshape(-1), torch.arange(len(choice)).repeat_interleave(K)])
        out = O.ds_point_conv(sd, '', x, (pos, pos[choice]), ei, mode == 'train')
    else:
        out = O.ds_point_conv(sd, '', x, pos, torch.stack([src, tgt]), mode == 'train')
    tag = 'dsconv/%s_%s' % (form, mode)
    close(out, g[tag + '/out'], OUT_TOL, 'out')
    (out * t(g[tag + '/gout'])).sum().backward()
    close(x.grad, g[tag + '/d_x'], GRAD_TOL, 'd_x')
    check_grads(sd, g, tag + '/grad')


def test_sparse_equals_dense_on_knn_graph():
    """A kNN graph without self loops run through the edge-list oracle == the dense oracle."""
    B, N, K, H = 1, 300, 9, 8
    pos = S.make_cloud(77, N)[None]
    nbr = torch.from_numpy(onative.oracle_knn_batch(pos, pos, K))
    z, y = t(S.uniform(77, 'z', (B, N, H))), t(S.uniform(77, 'y', (B, N, H)))
    c = torch.eye(H) + 0.1 * t(S.uniform(77, 'c', (H, H)))
    dense = O.crf_meanfield(z, y, nbr[:, :, 1:], c, 3)[0]
    tgt = torch.arange(N).repeat_interleave(K - 1)
    src = nbr[0, :, 1:].reshape(-1)
    sparse = O.sparse_crf_meanfield(z[0], y[0], tgt, src, c, 3)
    close(sparse, dense.numpy(), 1e-6, 'sparse vs dense')


# ----------------------------------------------------------------- callers either side of the network (8(f) rows 2-3)
def test_metrics_oracle_golden(golden):
    from oracle import eval_oracle as E
    g = golden('g9_eval.npz')
    yt, yp = g['m_yt'], g['m_yp']
    hist = sum(E.fast_hist(a, b, 13) for a, b in zip(yt, yp)) + E.fast_hist(yt[0], yp[1], 13)
    assert np.array_equal(hist, g['m_hist'])
    assert np.array_equal(E.fast_hist(yt, yp, 13, ignore_index=3), g['m_hist_ignore3'])
    sc, iu = E.scores(hist)
    assert np.allclose([sc[k] for k in sorted(sc)], g['m_scores'], rtol=1e-14, atol=0)
    assert np.allclose(iu, g['m_cls_iu'], rtol=1e-14, atol=0, equal_nan=True)
    # iou_from_confusions: absent classes take the mean of the present ones
    c = np.array([[5, 1, 0], [2, 7, 0], [0, 0, 0]], dtype=np.float64)
    iou = E.iou_from_confusions(c)
    assert abs(iou[2] - (iou[0] + iou[1]) / 2) < 1e-5 and abs(iou[0] - 5 / 8) < 1e-6


@pytest.mark.parametrize('split', ['train', 'test'])
def test_possibility_sampler_oracle_golden(golden, split):
    """The restated draw against Semantic3D._get_random run on the same clouds / possibilities / noise: crop
    membership, centred coordinates, labels and the possibility tables after six draws."""
    from oracle import eval_oracle as E
    g = golden('g9_eval.npz')
    clouds = [g['s_cloud0'], g['s_cloud1']]
    labels = [g['s_labels0'].astype(np.int64), g['s_labels1'].astype(np.int64)]
    poss = [g['s_poss0'].copy(), g['s_poss1'].copy()]
    weights = None if split == 'test' else [g['s_cw'][0][l - 1] for l in labels]
    minp = [float(p.min()) for p in poss]
    for draw in range(6):
        tag = 's_%s_%d_' % (split, draw)
        c = int(np.argmin(minp))
        assert c == int(g[tag + 'cloud'][0])
        idx, xyz, _ = E.possibility_draw(clouds[c], poss[c], 1500, g[tag + 'noise'],
                                         None if weights is None else weights[c])
        minp[c] = float(poss[c].min())
        ref_idx = g[tag + 'point_idx'].astype(np.int64)
        o, ro = np.argsort(idx), np.argsort(ref_idx)
        assert np.array_equal(idx[o], ref_idx[ro])
        assert np.array_equal(xyz[o], g[tag + 'pos'][ro])
        if split != 'test':
            assert np.array_equal(labels[c][idx[o]] - 1, g[tag + 'y'][ro])
        assert np.allclose(minp, g[tag + 'min_possibility'], rtol=1e-15, atol=0)
    for c in range(2):
        assert np.array_equal(poss[c], g['s_%s_possibility%d' % (split, c)])


@pytest.mark.parametrize('tag', ['T1_H64_G5', 'T3_H64_G5', 'T5_H16_G3', 'T2_H96_G2'])
def test_discrete_crf_oracle_golden(golden, tag):
    """oracle.discrete_crf against the reference's DiscreteCRFConv (ragged injected graph): q and every gradient."""
    g = golden('g10_discrete.npz')
    sd = params(sub(g, tag + '/sd'))
    logit, f = t(g['logit']).requires_grad_(True), t(g['f']).requires_grad_(True)
    tgt, src = t(g['tgt'].astype(np.int64)), t(g['src'].astype(np.int64))
    q = O.discrete_crf(sd, '', torch.softmax(logit, -1), f, tgt, src, int(tag[1]))
    close(q, g[tag + '/q'], OUT_TOL, 'q')
    (torch.log(q) * t(g[tag + '/gout'])).sum().backward()
    close(logit.grad, g[tag + '/d_logit'], GRAD_TOL, 'd_logit')
    close(f.grad, g[tag + '/d_f'], GRAD_TOL, 'd_f')
    check_grads(sd, g, tag + '/grad')


def test_shapenet_part_iou_oracle_golden(golden):
    from oracle import eval_oracle as E
    g = golden('g11_shapenet_score.npz')
    seg = {0: [0, 1, 2, 3], 4: [12, 13, 14, 15], 10: [30, 31, 32, 33, 34, 35], 15: [47, 48, 49], 8: [24, 25, 26, 27], 13: [41, 42, 43]}
    for i, c in enumerate(g['cats']):
        got = E.shapenet_part_iou(g['yt%d' % i].astype(np.int64), g['yp%d' % i].astype(np.int64), seg[int(c)])
        assert abs(got - g['ious'][i]) <= 1e-15
