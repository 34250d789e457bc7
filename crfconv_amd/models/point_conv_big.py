"""Dense PointConv encoder + CRF decoder on gfx950 kernels.

Drop-in for the reference network (models/point_conv_big.py:8-167): class names, constructor
arguments, the forward(data) contract and every state_dict key are the reference's; the gather /
weight-MLP / reduce chain of each convolution is one family of fused kernels
(crfconv_amd/csrc/pointconv.hip) instead of materialised [B, N, K, d] tensors."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..graph import table_of
from .common import MLP, Base, mlp_fork, mlp_group, mlp_join
from .continuous_crf_conv_big import ContinuousGaussianCRFConv as CRFConv

WIDTHS = (32, 64, 128, 256, 512)          # channel width of encoder level 0..4 (reference :113)


def _lrelu():
    return nn.LeakyReLU(negative_slope=0.1)


def _flat(t):
    return t.reshape(-1, t.shape[-1])


class PointConv(nn.Module):
    """out_i = sum_k weight_nn(p_i - p_j) * x_j, depth-wise (reference :8-58).

    `weight_nn` exists for its parameters (checkpoint layout: weight_nn.{0,1}.lin / .bn.batch_norm);
    the per-edge MLP is evaluated inside the kernels and never stored."""

    def __init__(self, d_model):
        super().__init__()
        self.weight_nn = nn.Sequential(MLP(3, d_model, activation=_lrelu()), MLP(d_model, d_model, activation=None))

    def _moments(self, table, p_src, p_tgt):
        """BatchNorm-1 statistics of rel = p_tgt[i] - p_src[j] (analytic in the moments), memoised per table: the two
        ResNet blocks of a level share them.  Positions edited in place since (jitter, a new batch in static buffers)
        are noticed through the tensors' version counters and the entry is recomputed into the same tensors."""
        key = ('moments', p_src.data_ptr(), p_tgt.data_ptr(), tuple(p_src.shape), tuple(p_tgt.shape))
        entry = table.cache.get(key)
        if entry is None:
            ps, pt = p_src.float().contiguous(), p_tgt.float().contiguous()
            keep_s = p_src if ps.data_ptr() == p_src.data_ptr() else ps      # version tracking needs the caller's tensor
            keep_t = p_tgt if pt.data_ptr() == p_tgt.data_ptr() else pt
            entry = table.cache[key] = ops.MomentsEntry(ops.relpos_moments(ps, pt, table), keep_s, keep_t)
        elif entry.stale():
            entry.refresh_(table)
        return entry

    def _geometry(self, pos, neighbor_idx):
        strided = not torch.is_tensor(pos)
        src = pos[0] if strided else pos
        p_src = _flat(src)
        p_tgt = _flat(pos[1]) if strided else p_src
        return table_of(neighbor_idx, src.shape[1]), p_src, p_tgt

    def prefold_entry(self, pos, neighbor_idx):
        """This layer's (W1, bn1, moments, table) for ops.point_conv_prefold."""
        table, p_src, p_tgt = self._geometry(pos, neighbor_idx)
        first = self.weight_nn[0]
        return first.lin.weight, first.bn.batch_norm, self._moments(table, p_src, p_tgt), table

    def forward(self, x, pos, neighbor_idx, prefold=None):
        table, p_src, p_tgt = self._geometry(pos, neighbor_idx)
        first, second = self.weight_nn[0], self.weight_nn[1]
        y = ops.point_conv(_flat(x), p_src, p_tgt, table, first.lin.weight, first.bn.batch_norm, second.lin.weight,
                           second.bn.batch_norm, self.training, moments=self._moments(table, p_src, p_tgt), prefold=prefold)
        return y.reshape(x.shape[0], -1, x.shape[-1])


class ResNetBBlock(nn.Module):
    """Bottleneck: lin_in -> PointConv -> lin_out, plus (max-pooled) shortcut (reference :61-88)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        mid = out_channels // 4
        self.lin_in = MLP(in_channels, mid, activation=_lrelu())
        self.lin_out = MLP(mid, out_channels, activation=None)
        self.shortcut = MLP(in_channels, out_channels, activation=None) if in_channels != out_channels else nn.Identity()
        self.point_conv = PointConv(mid)

    @staticmethod
    def max_pooling(x, idx):
        pooled = ops.neighbor_maxpool(_flat(x), table_of(idx, x.shape[1]))
        return pooled.reshape(x.shape[0], -1, x.shape[-1])

    def forward(self, x, pos, neighbor_idx, return_input_alias=False, prefold=None):
        """return_input_alias=True: also returns the block's input as the LAST alias of its fork chain (x -> lin_in -> strided
        shortcut): a further consumer of x -- the decoder stage that takes it as skip feature -- reads the alias, and its
        gradient is added inside this block's backward kernels instead of by an accumulation pass of autograd's."""
        strided = not torch.is_tensor(pos)
        skip = None
        sc = self.shortcut
        grouped = None
        if isinstance(sc, MLP):
            # lin_in and the shortcut read the same tensor: at the coarse levels ONE node runs both (two launches each way instead of four)
            grouped = mlp_group([(self.lin_in, x, True), (sc, x, False)], shared=True)
        if grouped is not None:
            (h_in, x), skip = grouped
            if strided:
                skip = self.max_pooling(skip, neighbor_idx)
            y = self.point_conv(h_in, pos, neighbor_idx, prefold=prefold)
            out = mlp_join(self.lin_out, y, skip, 0.01)
            return (out, x) if return_input_alias else out
        h_in, x = mlp_fork(self.lin_in, x)                 # x: now the alias whose gradient lin_in's backward adds to its own
        if (strided and self.training and isinstance(sc, MLP) and sc.bn is not None and sc.activation is None
                and sc.lin.bias is None and x.dtype == torch.float32 and sc.bn.batch_norm.affine):
            # shortcut MLP + max-pool as one node: BatchNorm applied while the pool gathers (ops.mlp_block_pool)
            pooled = ops.mlp_block_pool(_flat(x), sc.lin.weight, sc.bn.batch_norm, table_of(neighbor_idx, x.shape[1]),
                                        fork=return_input_alias)
            if pooled is not None:
                if return_input_alias:
                    pooled, alias = pooled
                    x = alias.reshape(x.shape)
                skip = pooled.reshape(x.shape[0], -1, pooled.shape[-1])
        if skip is None:
            if return_input_alias and isinstance(sc, MLP):
                skip, x = mlp_fork(sc, x)                  # coarse levels: the decoder's gradient joins inside the shortcut's dX product
            else:
                skip = sc(x)
            if strided:                                    # strided block: pool the shortcut onto the coarse points
                skip = self.max_pooling(skip, neighbor_idx)
        y = self.point_conv(h_in, pos, neighbor_idx, prefold=prefold)
        out = mlp_join(self.lin_out, y, skip, 0.01)        # lin_out + add + F.leaky_relu (default slope), as the reference
        return (out, x) if return_input_alias else out


class Upsampling(nn.Module):
    """Non-CRF decoder stage (reference :91-107): nearest up-sampling, MLP, fusion with the skip feature."""

    def __init__(self, down_channels, up_channels, out_channels):
        super().__init__()
        self.lin = MLP(down_channels, up_channels, activation=_lrelu())
        self.fusion = MLP(up_channels * 2, out_channels, activation=_lrelu())

    @staticmethod
    def upsampling(x, idx):
        up = ops.gather_rows(_flat(x), table_of(idx, x.shape[1]))
        return up.reshape(x.shape[0], -1, x.shape[-1])

    def forward(self, x_down, x_up, up_idx, neighbor_idx=None):
        return self.fusion(ops.cat2(x_up, self.lin(self.upsampling(x_down, up_idx))))


class PointConvResNet(Base):
    """Five encoder levels of two ResNet blocks (conv{l}_1 changes level / width, conv{l}_2 refines), four decoder
    stages deconv4..deconv1 (CRF mean-field layers, or plain Upsampling), per-point classifier (reference :110-167)."""

    def __init__(self, in_channels, n_classes, use_crf=True, steps=1):
        super().__init__()
        self.C = n_classes
        cin = in_channels
        for lvl, width in enumerate(WIDTHS, start=1):
            setattr(self, 'conv%d_1' % lvl, ResNetBBlock(cin, width))
            setattr(self, 'conv%d_2' % lvl, ResNetBBlock(width, width))
            cin = width
        for lvl in range(len(WIDTHS) - 1, 0, -1):          # deconv4 .. deconv1
            coarse, fine = WIDTHS[lvl], WIDTHS[lvl - 1]
            stage = CRFConv(coarse, fine, fine, steps=steps) if use_crf else Upsampling(coarse, fine, fine)
            setattr(self, 'deconv%d' % lvl, stage)
        self.classifier = nn.Sequential(MLP(WIDTHS[0], WIDTHS[0] * 4, activation=_lrelu()), nn.Dropout(p=0.5),
                                        nn.Linear(WIDTHS[0] * 4, n_classes))

    # optional callable(name), invoked while the step is being issued (and hence captured) where a phase begins: 'coarse' in front of
    # the forward's first level-2 block, 'coarse_backward' when the backward reaches the decoder's level 2 (a tensor hook).
    # data.CollatePipeline(gate=True).mark_on(name) makes one: the side stream's collate graph waits for that mark.
    phase_hook = None

    def forward(self, data):
        if self.training:
            from .. import train
            if train.autograph_wanted(data):                # the reference loop unwrapped: the step as two hipGraph replays (train.py)
                out = train.autograph_forward(self, data)
                if out is not None:
                    return out
            with ops.advance_counters(self):              # every BatchNorm below runs exactly once per forward
                return self._forward(data)
        return self._forward(data)

    def _forward(self, data):
        ms = data.multiscale
        # geometry of the ten encoder blocks: (block, positions, table indices)
        plan = [(self.conv1_1, ms[0].pos, ms[0].neighbor_idx), (self.conv1_2, ms[0].pos, ms[0].neighbor_idx)]
        for lvl in range(1, len(WIDTHS)):
            fine, coarse = ms[lvl - 1], ms[lvl]
            plan.append((getattr(self, 'conv%d_1' % (lvl + 1)), (fine.pos, coarse.pos), fine.sub_idx))
            plan.append((getattr(self, 'conv%d_2' % (lvl + 1)), coarse.pos, coarse.neighbor_idx))
        decoders = [getattr(self, 'deconv%d' % (lvl + 1)) for lvl in range(len(WIDTHS) - 2, -1, -1)]
        mats = [None] * len(decoders)
        if all(isinstance(d, CRFConv) for d in decoders):
            # (I + c^T c)^-1 of every CRF layer: ONE launch (and one back).  The matrices depend on parameters only, so the launch is
            # queued HERE: the first PointConv's statistics pass carries its workgroups (ops.crf_matrices_batched(ride=True)) and
            # the 27 us chain of dependent pivots leaves the forward's launch sequence; flush_riders() below covers eval mode
            mats = ops.crf_matrices_batched([d.c for d in decoders], ride=self.training and data.x.is_cuda)
        pre = [None] * len(plan)
        if self.training and data.x.is_cuda and not ops.state.no_prefold:      # BatchNorm-1 of all ten weight MLPs folded in ONE launch, up front
            pre = ops.point_conv_prefold([blk.point_conv.prefold_entry(p, i) for blk, p, i in plan], True)
        h = plan[0][0](data.x, plan[0][1], plan[0][2], prefold=pre[0])
        h = plan[1][0](h, plan[1][1], plan[1][2], prefold=pre[1])
        skips = [h]
        for lvl in range(1, len(WIDTHS)):
            if lvl == 2 and self.phase_hook is not None:
                self.phase_hook('coarse')                      # from here to the decoder's level 1 the launches are coarse-level ones
            (b1, p1, i1), (b2, p2, i2) = plan[2 * lvl], plan[2 * lvl + 1]
            # the level's output has three consumers (this block's lin_in and shortcut, the decoder): one fork chain, no add pass
            h, skips[-1] = b1(h, p1, i1, return_input_alias=True, prefold=pre[2 * lvl])
            h = b2(h, p2, i2, prefold=pre[2 * lvl + 1])
            skips.append(h)
        ops.flush_riders()
        for d, mat, lvl in zip(decoders, mats, range(len(WIDTHS) - 2, -1, -1)):
            if mat is not None:
                h = d(h, skips[lvl], ms[lvl].up_idx, ms[lvl].neighbor_idx, matrices=mat)
            else:
                h = d(h, skips[lvl], ms[lvl].up_idx, ms[lvl].neighbor_idx)
            if lvl == 2 and self.phase_hook is not None and h.requires_grad:
                # the gradient of the level-2 decoder output arrives when the backward leaves the fine levels: its coarse window begins
                h.register_hook(lambda g, f=self.phase_hook: (f('coarse_backward'), None)[1])
        head, drop, last = self.classifier[0], self.classifier[1], self.classifier[2]
        fused = logits = None
        if (self.training and type(drop) is nn.Dropout and not drop.inplace and isinstance(head, MLP) and head.bn is not None
                and head.lin.bias is None and isinstance(head.activation, nn.LeakyReLU) and head.bn.batch_norm.affine
                and h.dtype == torch.float32):
            # MLP -> Dropout -> Linear as ONE node (BatchNorm + LeakyReLU + a counter-based dropout mask in a single pass; the
            # mask's backward folded into the last Linear's input gradient), else MLP -> Dropout as one node
            slope = head.activation.negative_slope
            if type(last) is nn.Linear:
                logits = ops.mlp_dropout_linear(h, head.lin.weight, head.bn.batch_norm, slope, drop.p, last.weight, last.bias)
            if logits is None:
                fused = ops.mlp_block_dropout(h, head.lin.weight, head.bn.batch_norm, slope, drop.p)
        if logits is None:
            h = fused if fused is not None else drop(head(h))
            logits = ops.linear(h, last.weight, last.bias)
        h = logits
        return h.reshape(-1, self.C)
