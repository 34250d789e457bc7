"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the dense CRFConv / PointConv hot path.

A functional (no nn.Module) restatement, in plain PyTorch CPU ops, of what the
reference computes on the dense ``[B, N, C]`` + ``[B, N, K]`` path:

  * ``mlp``            <- models/common.py:26-40 (Linear(bias = not bn) -> FastBatchNorm1d -> act).
                          FastBatchNorm1d lives in torch_points3d (not vendored, version unpinned);
                          restated as BatchNorm1d(C, momentum=0.1, eps=1e-5) with statistics over
                          every leading dim.  Pinned by tests/golden (captured through a stub with
                          exactly that definition) -- "parity unpinned" at the third-party boundary.
  * ``crf_similarity`` <- models/continuous_crf_conv_big.py:49-54
  * ``crf_meanfield``  <- models/continuous_crf_conv_big.py:63-72
  * ``crf_conv``       <- models/continuous_crf_conv_big.py:56-78
  * ``point_conv``     <- models/point_conv_big.py:37-58
  * ``resnet_block``   <- models/point_conv_big.py:79-88 (+ max_pooling :74-77)
  * ``upsampling``     <- models/point_conv_big.py:97-107
  * ``pointconv_resnet`` <- models/point_conv_big.py:142-167
  * sparse twins ``sparse_crf_meanfield`` / ``guide_crf_conv`` / ``sparse_crf_conv`` /
    ``ds_point_conv`` <- models/continuous_crf_conv.py:50-69,112-133, models/point_conv.py:43-66
    (torch_geometric / torch_scatter are absent everywhere: "parity unpinned", pinned only by the
    dense == sparse equivalence test).

Parameters are read from a flat ``dict`` with the reference's state_dict key names, so a
state_dict captured from the reference drives this oracle directly.  Gradients come from torch
autograd on these CPU ops.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product (crfconv_amd) never does.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _rows(x, idx):
    """x [B, N, C], idx [B, M, K] (int64, per-cloud indices) -> [B, M, K, C]."""
    B, M, K = idx.shape
    C = x.shape[-1]
    flat = idx.reshape(B, M * K, 1).expand(B, M * K, C)
    return torch.gather(x, 1, flat).reshape(B, M, K, C)


def batch_norm(sd, prefix, x, training):
    """FastBatchNorm1d restated: statistics over all leading dims of x [..., C]."""
    w, b = sd[prefix + 'batch_norm.weight'], sd[prefix + 'batch_norm.bias']
    rm, rv = sd[prefix + 'batch_norm.running_mean'], sd[prefix + 'batch_norm.running_var']
    if x.dim() == 3:      # same layout the upstream module feeds BatchNorm1d: [B, C, N]
        return F.batch_norm(x.transpose(1, 2), rm, rv, w, b, training, BN_MOMENTUM, BN_EPS).transpose(1, 2)
    shape = x.shape
    y = F.batch_norm(x.reshape(-1, shape[-1]), rm, rv, w, b, training, BN_MOMENTUM, BN_EPS)
    return y.reshape(shape)


def mlp(sd, prefix, x, training, act_slope=None):
    """models/common.py:34-40. act_slope=None -> no activation."""
    x = F.linear(x, sd[prefix + 'lin.weight'], sd.get(prefix + 'lin.bias'))
    if (prefix + 'bn.batch_norm.weight') in sd:
        x = batch_norm(sd, prefix + 'bn.', x, training)
    if act_slope is not None:
        x = F.leaky_relu(x, act_slope)
    return x


# ------------------------------------------------------------------ CRF (dense)
def crf_similarity(y, nbr):
    """s[b,i,k] = softmax_k(-|y_i - y_j(i,k)|^2); y [B,N,H], nbr [B,N,Kn] -> [B,N,Kn]."""
    d = (y.unsqueeze(2) - _rows(y, nbr)).pow(2).sum(-1)
    return torch.softmax(-d, dim=2)


def crf_meanfield(z, y, nbr, c, steps):
    """x <- (z + (A x) C) (I + C)^-1 repeated `steps` times, x0 = z, C = c^T c."""
    s = crf_similarity(y, nbr)
    H = c.shape[0]
    C = c.t() @ c
    Minv = torch.linalg.inv(torch.eye(H, dtype=z.dtype) + C)
    x = z
    for _ in range(steps):
        m = (s.unsqueeze(-1) * _rows(x, nbr)).sum(2)
        x = (z + m @ C) @ Minv
    return x


def crf_conv(sd, prefix, unary, pairwise, up_idx, neighbor_idx, steps, training):
    nbr = neighbor_idx[:, :, 1:]                      # column 0 is taken to be the query itself
    x = mlp(sd, prefix + 'unary_nn.0.', unary, training, 0.1)
    x = mlp(sd, prefix + 'unary_nn.1.', x, training)
    y = mlp(sd, prefix + 'pairwise_nn.0.', pairwise, training, 0.1)
    y = mlp(sd, prefix + 'pairwise_nn.1.', y, training)
    z = _rows(x, up_idx)[:, :, 0, :]                  # nearest-coarse-point upsampling
    x = crf_meanfield(z, y, nbr, sd[prefix + 'c'], steps)
    x = mlp(sd, prefix + 'out_nn.', x, training, 0.1)
    return mlp(sd, prefix + 'fusion_nn.', torch.cat([x, pairwise], -1), training, 0.1)


# ------------------------------------------------------------------ PointConv (dense)
def point_conv(sd, prefix, x, pos, neighbor_idx, training):
    """Depth-wise conv: out[i,c] = sum_k w(p_i - p_j)[c] * x[j,c]."""
    if torch.is_tensor(pos):
        src, tgt = pos, pos
    else:
        src, tgt = pos
    B, M, K = neighbor_idx.shape
    rel = tgt.unsqueeze(2) - _rows(src, neighbor_idx)          # [B, M, K, 3]
    w = mlp(sd, prefix + 'weight_nn.0.', rel.reshape(B, M * K, 3), training, 0.1)
    w = mlp(sd, prefix + 'weight_nn.1.', w, training).reshape(B, M, K, -1)
    return (w * _rows(x, neighbor_idx)).sum(2)


def resnet_block(sd, prefix, x, pos, neighbor_idx, training):
    if (prefix + 'shortcut.lin.weight') in sd:
        res = mlp(sd, prefix + 'shortcut.', x, training)
    else:
        res = x
    if not torch.is_tensor(pos):
        res = _rows(res, neighbor_idx).max(2)[0]
    h = mlp(sd, prefix + 'lin_in.', x, training, 0.1)
    h = point_conv(sd, prefix + 'point_conv.', h, pos, neighbor_idx, training)
    h = mlp(sd, prefix + 'lin_out.', h, training)
    return F.leaky_relu(h + res)                     # default slope 0.01 (point_conv_big.py:88)


def upsampling(sd, prefix, x_down, x_up, up_idx, training):
    x_down = _rows(x_down, up_idx)[:, :, 0, :]
    x_down = mlp(sd, prefix + 'lin.', x_down, training, 0.1)
    return mlp(sd, prefix + 'fusion.', torch.cat([x_up, x_down], -1), training, 0.1)


def pointconv_resnet(sd, x, ms, steps, training, use_crf=True, dropout_mask=None):
    """ms: list of 5 dicts with pos / neighbor_idx / sub_idx / up_idx (level 4: pos, neighbor_idx).

    dropout_mask: optional [B, N, 128] 0/1 mask; Dropout(0.5) is applied as mask * 2 when
    training (the reference draws it from the global RNG, point_conv_big.py:138)."""
    def blk(name, h, pos, idx):
        return resnet_block(sd, name + '.', h, pos, idx, training)

    x1 = blk('conv1_1', x, ms[0]['pos'], ms[0]['neighbor_idx'])
    x1 = blk('conv1_2', x1, ms[0]['pos'], ms[0]['neighbor_idx'])
    feats = [x1]
    h = x1
    for lvl in range(1, 5):
        h = blk('conv%d_1' % (lvl + 1), h, (ms[lvl - 1]['pos'], ms[lvl]['pos']), ms[lvl - 1]['sub_idx'])
        h = blk('conv%d_2' % (lvl + 1), h, ms[lvl]['pos'], ms[lvl]['neighbor_idx'])
        feats.append(h)
    for lvl in (3, 2, 1, 0):
        name = 'deconv%d.' % (lvl + 1)
        if use_crf:
            h = crf_conv(sd, name, h, feats[lvl], ms[lvl]['up_idx'], ms[lvl]['neighbor_idx'], steps, training)
        else:
            h = upsampling(sd, name, h, feats[lvl], ms[lvl]['up_idx'], training)
    h = mlp(sd, 'classifier.0.', h, training, 0.1)
    if training:
        if dropout_mask is None:
            h = F.dropout(h, 0.5, True)
        else:
            h = h * dropout_mask * 2.0
    h = F.linear(h, sd['classifier.2.weight'], sd['classifier.2.bias'])
    return h.reshape(-1, h.shape[-1])


def training_loss(logits, labels, class_weights=None, ignore_index=-1):
    """trainval.py:101-104: y = data.y.reshape(-1) - 1; weighted CE with ignore_index."""
    y = labels.reshape(-1) - 1
    return F.cross_entropy(logits, y, weight=class_weights, ignore_index=ignore_index)


# ------------------------------------------------------------------ sparse twins
def segment_softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax restated: softmax of src grouped by index."""
    mx = torch.full((num_nodes,) + src.shape[1:], float('-inf'), dtype=src.dtype)
    mx = mx.scatter_reduce(0, index.reshape(-1, *[1] * (src.dim() - 1)).expand_as(src), src, 'amax')
    e = (src - mx[index]).exp()
    den = torch.zeros((num_nodes,) + src.shape[1:], dtype=src.dtype).index_add_(0, index, e)
    return e / (den[index] + 1e-16)


def sparse_crf_meanfield(z, y, tgt, src, c, steps):
    """Edge-list mean field (continuous_crf_conv.py:56-67 / 117-128).

    tgt[e] aggregates messages from src[e]; A is row-stochastic over each tgt's incoming edges."""
    N = z.shape[0]
    d = ((y[tgt] - y[src]) ** 2).sum(1, keepdim=True)
    s = segment_softmax(-d, tgt, N)
    H = c.shape[0]
    C = c.t() @ c
    Minv = torch.linalg.inv(torch.eye(H, dtype=z.dtype) + C)
    x = z
    for _ in range(steps):
        m = torch.zeros_like(z).index_add_(0, tgt, s * x[src])
        x = (z + m @ C) @ Minv
    return x


def _lin_bn(sd, prefix, x, training, i_lin=0, i_bn=1):
    x = F.linear(x, sd['%s%d.weight' % (prefix, i_lin)], sd.get('%s%d.bias' % (prefix, i_lin)))
    p = '%s%d.' % (prefix, i_bn)
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'],
                        sd[p + 'bias'], training, BN_MOMENTUM, BN_EPS)


def guide_crf_conv(sd, prefix, x, y, tgt, src, steps, training):
    """GuideGaussianCRFConv.forward (continuous_crf_conv.py:50-69) given its radius graph
    as (row = tgt, col = src)."""
    x = _lin_bn(sd, prefix + 'unary.', x, training)
    y = F.leaky_relu(_lin_bn(sd, prefix + 'pairwise.', y, training))
    x = sparse_crf_meanfield(x, y, tgt, src, sd[prefix + 'c'], steps)
    return F.leaky_relu(x)


def sparse_crf_conv(sd, prefix, x, y, edge_index, steps, training):
    """Sparse ContinuousGaussianCRFConv.forward (continuous_crf_conv.py:112-133):
    i = edge_index[0] aggregates from j = edge_index[1]."""
    i, j = edge_index
    xh = _lin_bn(sd, prefix + 'unary_net.', x, training)
    yh = _lin_bn(sd, prefix + 'pairwise_net.', y, training)
    xh = sparse_crf_meanfield(xh, yh, i, j, sd[prefix + 'c'], steps)
    xh = F.leaky_relu(_lin_bn(sd, prefix + 'mlp.', xh, training))
    return F.leaky_relu(_lin_bn(sd, prefix + 'fusion_net.', torch.cat([xh, y], -1), training))


def ds_point_conv(sd, prefix, x, pos, edge_index, training):
    """DepthwiseSeparablePointConv.forward (models/point_conv.py:43-66).

    edge_index = [col (source j); row (target i)] (PyG flow source_to_target)."""
    bipartite = not torch.is_tensor(pos)
    if not bipartite:
        n = pos.shape[0]
        keep = edge_index[0] != edge_index[1]
        loops = torch.arange(n, dtype=edge_index.dtype)
        edge_index = torch.cat([edge_index[:, keep], torch.stack([loops, loops])], 1)
        pos_src, pos_dst = pos, pos
    else:
        pos_src, pos_dst = pos
    col, row = edge_index
    n_dst = pos_dst.shape[0]
    res = x
    if bipartite:
        C = x.shape[1]
        res = torch.full((n_dst, C), float('-inf'), dtype=x.dtype).scatter_reduce(
            0, row.unsqueeze(1).expand(-1, C), x[col], 'amax')
    if (prefix + 'mlp4.0.weight') in sd:
        res = _lin_bn(sd, prefix + 'mlp4.', res, training)
    h = F.leaky_relu(_lin_bn(sd, prefix + 'mlp2.', x, training))
    w = F.leaky_relu(_lin_bn(sd, prefix + 'mlp1.', pos_dst[row] - pos_src[col], training))
    w = _lin_bn(sd, prefix + 'mlp1.', w, training, 3, 4)
    msg = torch.zeros((n_dst, h.shape[1]), dtype=h.dtype).index_add_(0, row, w * h[col])
    out = _lin_bn(sd, prefix + 'mlp3.', msg, training)
    return F.leaky_relu(out + res)


def discrete_crf(sd, prefix, p, f, tgt, src, steps):
    """models/discrete_crf_conv.py:40-63 on an explicit edge list (tgt = ``row``, src = ``col`` of the radius graph,
    :44): Gaussian-kernel edge weights in the hidden spaces f F_g (:49-54), then ``steps`` rounds of
    message passing (scatter_add of w * q[col] at row, :58), label compatibility (:59) and soft-max (:60).
    Parameters ``F`` [G, D, H], ``W`` [G, 1], ``C`` [L, L].  Pinned by g10_discrete.npz (the reference layer run
    with the graph injected for the absent torch_cluster radius search; torch_scatter.scatter_add restated as
    index_add -- "parity unpinned" at that third-party boundary)."""
    Fk, Wk, C = sd[prefix + 'F'], sd[prefix + 'W'], sd[prefix + 'C']
    n = p.shape[0]
    u = -torch.log(p)
    fk = torch.einsum('nd,gdh->ngh', f, Fk)                       # [N, G, H]
    diff = fk[src] - fk[tgt]
    w = torch.exp(-(diff ** 2).sum(-1)) @ Wk                      # [E, 1]
    q = p
    for _ in range(steps):
        msg = torch.zeros((n, p.shape[1]), dtype=p.dtype).index_add_(0, tgt, q[src] * w)
        q = torch.softmax(-u - msg @ C, dim=-1)
    return q
