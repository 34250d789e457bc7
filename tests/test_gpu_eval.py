"""-m gpu: the callers either side of the network through libcrfconv_amd.so against the
oracle (oracle/eval_oracle.py) and the fixtures captured from the reference (g9_eval.npz).
Integer work bit-exact; the float32 vote update bit-exact on given probabilities, 1e-6 with the fused soft-max."""
import numpy as np
import pytest
import torch

import _seeded as S
from gpu_util import DEV, t
from oracle import eval_oracle as E

pytestmark = pytest.mark.gpu


def test_running_score_golden(golden):
    from crfconv_amd.utils import runningScore
    g = golden('g9_eval.npz')
    yt, yp = t(g['m_yt']), t(g['m_yp'])
    rs = runningScore(13, ignore_index=-1)
    rs.update(yt, yp)
    rs.update(yt[0], yp[1])
    assert np.array_equal(rs.confusion_matrix, g['m_hist'])
    sc, cls_iu = rs.get_scores()
    assert list(sorted(sc)) == list(g['m_score_names'])
    assert np.allclose([sc[k] for k in sorted(sc)], g['m_scores'], rtol=1e-14, atol=0)
    assert np.allclose([cls_iu[c] for c in range(13)], g['m_cls_iu'], rtol=1e-14, atol=0, equal_nan=True)
    rs.reset()
    assert rs.confusion_matrix.sum() == 0
    rs3 = runningScore(13, ignore_index=3)
    rs3.update(yt, yp)
    assert np.array_equal(rs3.confusion_matrix, g['m_hist_ignore3'])


@pytest.mark.parametrize('n_cls,n', [(13, 163840), (2, 1000), (100, 50000), (8, 1)])
def test_confusion_from_logits_vs_oracle(n_cls, n):
    from crfconv_amd.utils import runningScore
    from crfconv_amd.utils.metrics import iou_from_confusions
    rng = np.random.default_rng(n_cls)
    logits = rng.standard_normal((n, n_cls)).astype(np.float32)
    logits[::7, 1] = logits[::7, 0] = 9.0                     # ties: first maximum wins
    y = rng.integers(0, n_cls + 2, n)                         # raw labels 0 .. n_cls+1; class = y - 1
    rs = runningScore(n_cls, ignore_index=-1)
    rs.update_from_logits(t(y), t(logits), label_shift=1)
    want = E.fast_hist(y - 1, logits.argmax(1), n_cls)
    assert np.array_equal(rs.confusion_matrix, want)
    assert np.allclose(iou_from_confusions(want), E.iou_from_confusions(want), rtol=1e-12)
    bad = runningScore(n_cls)
    bad.update(t(np.zeros(4, np.int64)), t(np.full(4, n_cls, np.int64)))
    with pytest.raises(ValueError):
        bad.get_scores()


def test_vote_accumulator_vs_oracle():
    from crfconv_amd.sampling import VoteAccumulator
    rng = np.random.default_rng(5)
    sizes, C, B, N = [5000, 3000], 8, 4, 1024
    acc = VoteAccumulator(sizes, C, smooth=0.98)
    ref = [np.zeros((n, C), np.float32) for n in sizes]
    fused = VoteAccumulator(sizes, C, smooth=0.98)
    for it in range(5):
        clouds = rng.integers(0, 2, B)
        pidx = np.stack([rng.permutation(sizes[c])[:N] for c in clouds])
        logits = (3 * rng.standard_normal((B * N, C))).astype(np.float32)
        probs = E.softmax32(logits)
        acc.update(t(pidx), t(clouds.reshape(B, 1)), probs=t(probs))
        fused.update(t(pidx), t(clouds.reshape(B, 1)), logits=t(logits))
        for b in range(B):
            E.vote_update(ref[clouds[b]], pidx[b], probs.reshape(B, N, C)[b], 0.98)
    acc.check()
    for c in range(2):
        assert np.array_equal(acc.test_probs[c].cpu().numpy(), ref[c])              # same float32 op sequence
        assert np.abs(fused.test_probs[c].cpu().numpy() - ref[c]).max() < 1e-6      # device soft-max inside
        proj = rng.integers(0, sizes[c], 20000)
        assert np.array_equal(acc.project(c, t(proj)).cpu().numpy(), E.vote_project(ref[c], proj))
    acc.update(t(np.full((1, 4), 10 ** 6)), t(np.zeros((1, 1), np.int64)), probs=t(np.zeros((4, C), np.float32)))
    with pytest.raises(IndexError):
        acc.check()


@pytest.mark.parametrize('split', ['train', 'test'])
def test_possibility_sampler_golden(golden, split):
    """Six consecutive draws against Semantic3D._get_random (fixture): cloud choice, crop membership, centred
    float32 coordinates, labels, colours and the float64 possibility tables -- all bit-exact."""
    from crfconv_amd.sampling import PossibilitySampler
    g = golden('g9_eval.npz')
    clouds = [t(g['s_cloud0']), t(g['s_cloud1'])]
    labels = [t(g['s_labels0'].astype(np.int64)), t(g['s_labels1'].astype(np.int64))]
    rgb = [t(g['s_rgb0']), t(g['s_rgb1'])]
    smp = PossibilitySampler(clouds, rgb=rgb, labels=labels, num_points=1500, class_weight=g['s_cw'][0],
                             label_to_idx={l: i for i, l in enumerate(range(1, 9))}, split=split,
                             possibility=[g['s_poss0'], g['s_poss1']])
    for draw in range(6):
        tag = 's_%s_%d_' % (split, draw)
        d = smp.get_random(noise=g[tag + 'noise'], perm=False)
        assert int(d.cloud_idx.item()) == int(g[tag + 'cloud'][0])
        idx, ref_idx = d.point_idx.cpu().numpy(), g[tag + 'point_idx'].astype(np.int64)
        o, ro = np.argsort(idx), np.argsort(ref_idx)
        assert np.array_equal(idx[o], ref_idx[ro])
        assert np.array_equal(d.pos.cpu().numpy()[o], g[tag + 'pos'][ro])
        assert np.array_equal(d.rgb.cpu().numpy()[o], g[tag + 'rgb'][ro])
        assert np.array_equal(d.y.cpu().numpy()[o], g[tag + 'y'][ro])
        assert np.array_equal(smp.min_possibility, g[tag + 'min_possibility'])
    for c in range(2):
        assert np.array_equal(smp.possibility[c].cpu().numpy(), g['s_%s_possibility%d' % (split, c)])


def test_possibility_sampler_full_size_properties():
    """Config-5-sized cloud (1 M points, crops of 65536): a crop is exactly the ball of its k nearest points, the
    shuffle is a permutation of it, possibilities only grow and the seed's own possibility grows by its full weight."""
    from crfconv_amd.sampling import PossibilitySampler
    gen = torch.Generator().manual_seed(3)
    pts = (torch.rand(1 << 20, 3, generator=gen) * torch.tensor([60.0, 60.0, 15.0])).to(DEV)
    smp = PossibilitySampler([pts], num_points=65536, split='test', generator=gen)
    before = smp.possibility[0].clone()
    d = smp.get_random()
    idx = d.point_idx
    assert idx.unique().numel() == 65536
    c = d.center
    d2 = ((pts.double() - c) ** 2).sum(1)
    inside = torch.zeros(pts.shape[0], dtype=torch.bool, device=DEV)
    inside[idx] = True
    assert float(d2[inside].max()) <= float(d2[~inside].min())
    delta = smp.possibility[0] - before
    assert float(delta.min()) >= 0 and float(delta[~inside].abs().max()) == 0
    assert float(delta.max()) <= 1.0 + 1e-12
    assert torch.equal(d.pos[:, 2], pts[idx, 2])
    d_b = smp.get_random()
    assert not torch.equal(d_b.center, d.center)


def test_running_score_shapenet_golden(golden):
    """runningScoreShapeNet on the device confusion kernel against the reference class (fixture): per-shape IoU,
    instance mean, category table (categories never seen stay nan, as there)."""
    from crfconv_amd.utils import runningScoreShapeNet
    g = golden('g11_shapenet_score.npz')
    rs = runningScoreShapeNet()
    for i, c in enumerate(g['cats']):
        iou = rs.update(t(g['yt%d' % i].astype(np.int64)), t(g['yp%d' % i].astype(np.int64)), int(c))
        assert abs(iou - g['ious'][i]) <= 1e-12
    p, mp, cls = rs.get_scores()
    assert abs(p - float(g['pIoU'])) <= 1e-6
    assert np.isnan(mp) and np.isnan(float(g['mpIoU']))
    assert np.allclose([cls[k] for k in sorted(cls)], g['cls'], rtol=1e-6, equal_nan=True)
    assert list(sorted(cls)) == list(g['cls_names'])


def _scene(n, seed):
    rng = np.random.default_rng(seed)
    pts = (rng.random((n, 3)) * np.array([6.0, 6.0, 2.0])).astype(np.float32)
    rgb = rng.random((n, 3)).astype(np.float32)
    return pts, rgb


@pytest.mark.parametrize('graphed', [False, True])
def test_tiled_scene_pipeline_against_the_oracle_chain(graphed):
    """Config 5 / 3 as a PIPELINE on a small scene: possibility sampler -> multiscale_compute -> PointConvBig (eval) ->
    soft-max votes -> re-projection (trainval.py:170-203, datasets/semantic3d_dataset.py:423-460), `sampling.vote_scene` against the
    oracle chain on the SAME draws: oracle/eval_oracle.py possibility_draw (crop membership and possibilities bit-exact), oracle/crf_oracle.py
    network on the device collate's tables (kNN parity is test_gpu_native's), vote_update / vote_project.  Vote tables within 1e-5, final
    per-point labels equal wherever the two best votes are not a near-tie.  graphed: collate and forward of the crops after the first as
    hipGraph replays (static batch refreshed in place per crop)."""
    from crfconv_amd import models
    from crfconv_amd.sampling import PossibilitySampler, VoteAccumulator, vote_scene
    from oracle import crf_oracle as O
    n, crop, n_crops, C, K, T = 24000, 6000, 4, 8, 16, 3
    pts, rgb = _scene(n, 11)
    net = models.PointConvBig(6, C, use_crf=True, steps=T)
    sd = S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 9)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    poss0 = (np.random.default_rng(5).standard_normal(n) * 1e-3)
    smp = PossibilitySampler([t(pts)], rgb=[t(rgb)], num_points=crop, split='test', generator=torch.Generator().manual_seed(77), possibility=[poss0])
    votes = VoteAccumulator([n], C, device=DEV)
    # the oracle side replays the same draws: the sampler takes randn(3) * noise_scale and randperm(k) from its generator, in that order
    gen = torch.Generator().manual_seed(77)
    poss = poss0.astype(np.float64).copy()
    ref_votes = np.zeros((n, C), np.float32)
    prm = {k: v.clone() for k, v in sd.items()}
    crops = []

    def keep(data, logits, point_idx):           # (graphed: `data` is the static batch, overwritten by the next crop)
        crops.append((data.x.cpu().clone(), [{k: getattr(l, k).cpu().clone() for k in ('pos', 'neighbor_idx', 'sub_idx', 'up_idx')} for l in data.multiscale],
                      point_idx.reshape(-1).cpu().numpy().copy()))
    vote_scene(smp, net, votes, n_crops, kernel_size=(K,) * 5, generator=torch.Generator().manual_seed(3), graphed=graphed, on_crop=keep)
    votes.check()
    assert len(crops) == n_crops
    for x_c, ms, got_idx in crops:
        noise = (torch.randn(3, dtype=torch.float64, generator=gen) * smp.noise_scale).numpy()
        torch.randperm(crop, generator=gen)                                   # (the shuffle: order inside the crop, irrelevant to the votes)
        order, xyz, _ = E.possibility_draw(pts, poss, crop, noise)
        assert np.array_equal(np.sort(got_idx), np.sort(order))               # the same crop
        with torch.no_grad():
            ref_logits = O.pointconv_resnet(prm, x_c, ms, T, False, True)
        E.vote_update(ref_votes, got_idx, E.softmax32(ref_logits.numpy()), 0.98)
    assert np.array_equal(smp.possibility[0].cpu().numpy(), poss)             # float64 possibilities: bit-exact
    got = votes.test_probs[0].cpu().numpy()
    assert np.abs(got - ref_votes).max() < 1e-5, np.abs(got - ref_votes).max()
    proj = np.random.default_rng(2).integers(0, n, 50000)
    labels = votes.project(0, t(proj)).cpu().numpy()
    ref_labels = E.vote_project(ref_votes, proj)
    top2 = np.sort(ref_votes[proj], axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-5
    assert clear.mean() > 0.5 and np.array_equal(labels[clear], ref_labels[clear])
    visited = ref_votes.sum(1) > 0
    assert 0.3 < visited.mean() <= 1.0


def test_votes_merged_over_two_accumulators_equal_one_accumulator_in_rank_order():
    """Crops sharded over ranks: rank r votes the crops i % world == r into its own tables (visit counts beside them),
    `fold_` / `merge` combine them in rank order.  The result must equal ONE accumulator that applied rank 0's crops first, then rank 1's
    (the reference's update is a running mean, trainval.py:188-189: order-dependent; the merge fixes the order, not the rule)."""
    from crfconv_amd.sampling import VoteAccumulator
    n, C, k = 5000, 8, 1500
    rng = np.random.default_rng(4)
    crops = [(rng.choice(n, k, replace=False), rng.standard_normal((k, C)).astype(np.float32)) for _ in range(7)]
    parts = [VoteAccumulator([n], C, device=DEV, track_visits=True) for _ in range(2)]
    for i, (idx, logits) in enumerate(crops):
        parts[i % 2].update(t(idx).reshape(1, -1), torch.tensor([[0]], device=DEV), logits=t(logits))
    one = VoteAccumulator([n], C, device=DEV, track_visits=True)
    for r in range(2):
        for i, (idx, logits) in enumerate(crops):
            if i % 2 == r:
                one.update(t(idx).reshape(1, -1), torch.tensor([[0]], device=DEV), logits=t(logits))
    parts[0].fold_(parts[1].test_probs, parts[1].visits)
    assert torch.equal(parts[0].visits[0], one.visits[0]) and int(one.visits[0].sum()) == 7 * k
    err = float((parts[0].test_probs[0] - one.test_probs[0]).abs().max())
    assert err < 2e-7, err
    proj = t(rng.integers(0, n, 20000))
    a, b = parts[0].project(0, proj), one.project(0, proj)
    assert float((a == b).float().mean()) > 0.9999
