"""Batch containers with the fields the reference model reads (torch_geometric ``Data`` /
torch_points3d ``MultiScaleData`` as used in datasets/semantic3d_dataset.py:527-534) and the
multiscale collate (``_multiscale_compute_fn``, :501-534) run on the GPU with the HIP kNN."""
import torch

from .utils import nearest_neighbors


class Data:
    """Attribute bag of tensors with ``.to(device)`` (the part of torch_geometric.data.Data the
    hot path touches: trainval.py:98 ``data.to(device)``)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k in self.__dict__ if not k.startswith('_')]

    def _apply(self, fn):
        def go(v):
            if torch.is_tensor(v):
                return fn(v)
            if isinstance(v, Data):
                return v._apply(fn)
            if isinstance(v, (list, tuple)):
                return type(v)(go(u) for u in v)
            return v
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            setattr(out, k, go(v))
        return out

    def to(self, device, non_blocking=False):
        return self._apply(lambda t: t.to(device, non_blocking=non_blocking))

    def cuda(self):
        return self.to('cuda')

    def cpu(self):
        return self.to('cpu')

    def __repr__(self):
        parts = []
        for k in self.keys:
            v = getattr(self, k)
            parts.append('%s=%s' % (k, list(v.shape) if torch.is_tensor(v) else type(v).__name__))
        return '%s(%s)' % (self.__class__.__name__, ', '.join(parts))


class MultiScaleData(Data):
    """x [B,N,C], y [B,N], point_idx, cloud_idx, multiscale = [Data(pos, neighbor_idx, sub_idx, up_idx)]."""

    def __init__(self, x=None, y=None, point_idx=None, cloud_idx=None, multiscale=None, **kwargs):
        super().__init__(x=x, y=y, point_idx=point_idx, cloud_idx=cloud_idx, multiscale=multiscale, **kwargs)

    def load_(self, other, defer_check=False):
        """Copies another batch of the SAME shapes into this batch's tensors (in place) and refreshes the neighbour
        tables / reverse CSRs / rel-pos moments derived from them into their existing buffers.  This batch's tensors
        are thereby static buffers: a hipGraph captured on a training step over ``self`` trains on ``other`` at its
        next replay.  (All table sizes are fixed for fixed B, N, K and ratios.)  defer_check: the tables' range check without its
        host synchronisation (one per table otherwise): graph.check_pending() -- run by the next load_ -- raises instead."""
        from .graph import table_of

        pairs, tables = [], []

        def copy(dst, src, what):
            if dst is None or src is None:
                if dst is not src:
                    raise ValueError('load_: %s present in one batch only' % what)
                return
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError('load_: %s is %s %s here, %s %s there' % (what, tuple(dst.shape), dst.dtype,
                                                                           tuple(src.shape), src.dtype))
            if dst.is_cuda and src.is_cuda and dst.is_contiguous() and src.is_contiguous():
                pairs.append((dst, src))                   # all of them in ONE launch below
            else:
                dst.copy_(src)
        for name in ('x', 'y', 'point_idx', 'cloud_idx', 'order'):
            a, b = getattr(self, name, None), getattr(other, name, None)
            if torch.is_tensor(a) or torch.is_tensor(b):
                copy(a, b, name)
        if len(self.multiscale) != len(other.multiscale):
            raise ValueError('load_: different number of scales')
        for i, (mine, theirs) in enumerate(zip(self.multiscale, other.multiscale)):
            copy(mine.pos, theirs.pos, 'multiscale[%d].pos' % i)
        for i, (mine, theirs) in enumerate(zip(self.multiscale, other.multiscale)):
            n_here = mine.pos.shape[1]
            n_next = self.multiscale[i + 1].pos.shape[1] if i + 1 < len(self.multiscale) else None
            for name, n_src in (('neighbor_idx', n_here), ('sub_idx', n_here), ('up_idx', n_next)):
                a, b = getattr(mine, name, None), getattr(theirs, name, None)
                if not (torch.is_tensor(a) or torch.is_tensor(b)):
                    continue
                copy(a, b, 'multiscale[%d].%s' % (i, name))
                if getattr(a, '_crf_tables', None):          # a table was derived from it: same buffers, new content
                    tables.extend((a, key[0]) for key in list(a._crf_tables))
        if pairs:
            import ctypes
            from . import _lib
            from .graph import stream_ptr
            jobs = (_lib.CopyJob * len(pairs))(*[_lib.CopyJob(src.data_ptr(), dst.data_ptr(), dst.numel() * dst.element_size())
                                                 for dst, src in pairs])
            _lib.call('crfconv_copy_jobs', ctypes.cast(jobs, ctypes.c_void_p), len(pairs), stream_ptr())
            for dst, _ in pairs:                           # written by a custom kernel: the version counters (table / moments
                torch.autograd.graph.increment_version(dst)    # memos compare them) must say so
        from .graph import batched_reverse
        with batched_reverse(defer_check=defer_check):     # every table's reverse CSR in one set of launches at the end
            for a, n_src in tables:                        # after the copies: the refreshes read the new content
                table_of(a, n_src)
        return self


def _spread3(v):
    """Spread the low 10 bits of v so that there are two zero bits between consecutive bits."""
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton_codes(pos):
    """30-bit Morton (Z-order) codes [B, N] int64 of pos [B, N, 3]: ten bits per axis of the position inside the cloud's
    bounding cube.  On the device: two launches (csrc/knn.hip: crfconv_morton_codes)."""
    if pos.is_cuda and pos.dtype == torch.float32 and pos.dim() == 3 and pos.shape[-1] == 3 and pos.shape[1] > 0:
        from . import _lib
        from .graph import ptr, stream_ptr
        p = pos.detach().contiguous()
        B, N, _ = p.shape
        box = torch.empty((B, 4), dtype=torch.float32, device=p.device)
        code = torch.empty((B, N), dtype=torch.int64, device=p.device)
        _lib.call('crfconv_morton_codes', ptr(p), B, N, ptr(box), ptr(code), stream_ptr())
        return code
    lo = pos.amin(dim=1, keepdim=True)                     # host tensors (tests, tools): the same arithmetic, op by op
    ext = (pos.amax(dim=1, keepdim=True) - lo).amax(dim=2, keepdim=True).clamp_min(1e-20)
    q = ((pos - lo) / ext * 1023.0).clamp_(0, 1023).to(torch.int64)
    return _spread3(q[..., 0]) | (_spread3(q[..., 1]) << 1) | (_spread3(q[..., 2]) << 2)


def morton_order(pos, out=None):
    """Per-cloud permutation that sorts points along a 30-bit Morton (Z-order) curve.  pos [B, N, 3].  On the device the stable
    argsort is this library's own (csrc/collate.hip: crfconv_argsort_codes -- bit-identical to torch.argsort(stable=True),
    no scratch memory, so it can sit inside a captured graph); `out` [B, N] int64 receives it when given."""
    code = morton_codes(pos)
    if code.is_cuda and code.dim() == 2 and code.shape[1] > 0:
        from . import _lib
        from .graph import ptr, stream_ptr
        B, N = code.shape
        order = out if out is not None else torch.empty((B, N), dtype=torch.int64, device=code.device)
        nbytes = _lib.load().crfconv_argsort_codes_workspace(B, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=code.device)
        _lib.call('crfconv_argsort_codes', ptr(code), B, N, ptr(order), ptr(ws), nbytes, stream_ptr())
        return order
    order = torch.argsort(code, dim=1, stable=True)
    if out is not None:
        out.copy_(order)
        return out
    return order


def random_subsets_device(sizes, counts, seed, counter, outs, ranks=None):
    """outs[l] [counts[l]] int64 (device) <- a uniformly random subset of range(sizes[l]) in ascending order, one launch for all
    levels (csrc/collate.hip: crfconv_random_subsets).  A function of (seed, counter[0], l): `counter` is a one-element int64
    DEVICE tensor the caller advances per batch.  Not torch.randperm's draws.  ranks[l] [sizes[l]] int32 (optional): the membership
    table of the subset (position of a point in it, -1 outside) for up_index_from_table."""
    import ctypes
    from . import _lib
    from .graph import stream_ptr
    L = len(sizes)
    n = (ctypes.c_int * L)(*[int(v) for v in sizes])
    s = (ctypes.c_int * L)(*[int(v) for v in counts])
    o = (ctypes.c_void_p * L)(*[t.data_ptr() for t in outs])
    for t, c in zip(outs, counts):
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() >= c):
            raise _lib.CrfConvError('random_subsets_device: outputs must be contiguous int64 device tensors of the subset sizes')
    r = None
    if ranks is not None:
        for t, c in zip(ranks, sizes):
            if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and t.numel() >= c):
                raise _lib.CrfConvError('random_subsets_device: ranks must be contiguous int32 device tensors of the level sizes')
        r = ctypes.cast((ctypes.c_void_p * L)(*[t.data_ptr() for t in ranks]), ctypes.c_void_p)
    _lib.call('crfconv_random_subsets', ctypes.cast(n, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
              ctypes.cast(o, ctypes.c_void_p), r, L, int(seed) & 0xFFFFFFFFFFFFFFFF, counter.data_ptr(), stream_ptr())


_UP_WS = {}


def up_index_from_table(pos, neighbor_idx, choice, rank=None, sub_pos=None):
    """up_idx [B, N, 1] int64 = knn_batch(pos[:, choice], pos, 1) (datasets/semantic3d_dataset.py:524), bit-identical, from the level's own
    K-nearest table (csrc/collate.hip: crfconv_upindex_from_table).  choice [S] int64: the subset shared by the clouds; rank [N] int32:
    its membership table, sub_pos [B, S, 3] its positions (both built here when the caller has none)."""
    from . import _lib
    from .graph import ptr, stream_ptr
    B, N, K = neighbor_idx.shape
    S = choice.numel()
    dev = pos.device
    if rank is None:
        rank = torch.full((N,), -1, dtype=torch.int32, device=dev)
        rank[choice] = torch.arange(S, dtype=torch.int32, device=dev)
    pos = pos.detach().to(torch.float32).contiguous()
    sub_pos = pos[:, choice].contiguous() if sub_pos is None else sub_pos.detach().to(torch.float32).contiguous()
    out = torch.empty((B, N, 1), dtype=torch.int64, device=dev)
    nbytes = _lib.load().crfconv_upindex_workspace(B, N)
    # per (device, shape) workspace: its state words are zero between uses (never shared by launches in flight on two streams)
    capturing = torch.cuda.is_current_stream_capturing()
    key = (dev.index, 'capture' if capturing else int(torch.cuda.current_stream(dev).cuda_stream), B, N)
    ws = _UP_WS.get(key)
    if ws is None:
        ws = _UP_WS[key] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        if not capturing and (dev.index, 'capture', B, N) not in _UP_WS:      # the eager warm-up also makes the buffer captured launches
            _UP_WS[(dev.index, 'capture', B, N)] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)      # use: a capture never allocates
    _lib.call('crfconv_upindex_from_table', ptr(pos), ptr(sub_pos), ptr(neighbor_idx.contiguous()), ptr(rank), B, N, K, S, ptr(out), ptr(ws), nbytes,
              stream_ptr())
    return out


def _fps_choice(pos, n_sample):
    """[B, n_sample] per-cloud farthest-point picks in pick order, first pick = point 0 -- what the reference's
    ``tpcuda.furthest_point_sampling(pos, n)`` returns (datasets/semantic3d_dataset.py:520) -- on csrc/knn.hip: fps_kernel
    (one workgroup per cloud; ties -> lower index)."""
    from . import _lib
    from .graph import ptr, stream_ptr
    B, N, _ = pos.shape
    dev = pos.device
    ar = torch.arange(B, dtype=torch.int64, device=dev)
    starts, cnt = ar * N, torch.full((B,), N, dtype=torch.int64, device=dev)
    ostart, npick = ar * n_sample, torch.full((B,), n_sample, dtype=torch.int64, device=dev)
    first = torch.zeros(B, dtype=torch.int64, device=dev)
    p = pos.detach().to(torch.float32).contiguous().reshape(B * N, 3)
    out = torch.empty(B * n_sample, dtype=torch.int64, device=dev)
    ws = torch.empty(B * N, dtype=torch.float32, device=dev)
    _lib.call('crfconv_fps', ptr(p), B, ptr(starts), ptr(cnt), ptr(ostart), ptr(npick), ptr(first), ptr(ws), ptr(out),
              stream_ptr())
    return out.reshape(B, n_sample) - starts[:, None]          # the kernel reports global rows


def pick_rows(tensors, index, per_cloud):
    """[t[:, index] for t in tensors] (index [S] shared by all clouds) or per-cloud picks (index [B, S]) for [B, N, ...]
    CUDA tensors, as ONE library launch per 8 tensors (crfconv_gather_rows_batched); None entries pass through."""
    import ctypes
    from . import _lib
    from .graph import ptr, stream_ptr
    out = [None] * len(tensors)
    live = [(i, t.contiguous()) for i, t in enumerate(tensors) if t is not None]
    if not live:
        return out
    index = index.contiguous()
    B, N = live[0][1].shape[:2]
    S = index.shape[-1]
    ok = all(t.is_cuda and t.dim() >= 2 and t.shape[0] == B and t.shape[1] == N and (t[0, 0].numel() * t.element_size()) % 4 == 0
             and t.numel() > 0 for _, t in live) and index.is_cuda and index.dtype == torch.int64 and S > 0
    if not ok:
        for i, t in live:
            if per_cloud:
                idx = index.reshape(index.shape + (1,) * (t.dim() - 2)).expand(index.shape + t.shape[2:])
                out[i] = torch.gather(t, 1, idx)
            else:
                out[i] = t[:, index].contiguous()
        return out
    for o in range(0, len(live), 8):
        part = live[o:o + 8]
        dsts = [torch.empty((B, S) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device) for _, t in part]
        n = len(part)
        src = (ctypes.c_void_p * n)(*[t.data_ptr() for _, t in part])
        dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
        rb = (ctypes.c_int * n)(*[t[0, 0].numel() * t.element_size() for _, t in part])
        _lib.call('crfconv_gather_rows_batched', ctypes.cast(src, ctypes.c_void_p), ctypes.cast(dst, ctypes.c_void_p),
                  ctypes.cast(rb, ctypes.c_void_p), n, ptr(index), 1 if per_cloud else 0, B, N, S, stream_ptr())
        for (i, _), d in zip(part, dsts):
            out[i] = d
    return out


def multiscale_compute(pos, x=None, y=None, point_idx=None, cloud_idx=None, kernel_size=(16, 16, 16, 16, 16),
                       ratio=(4, 4, 4, 4, 2), num_scales=5, generator=None, choices=None, sort=None,
                       sample_method='random', order=None, ranks=None):
    """The reference collate on the device (datasets/semantic3d_dataset.py:512-528):
    per scale  neighbor_idx = knn(pos, pos, K);  one random subset shared by all clouds;
    sub_idx = neighbor_idx[:, choice];  up_idx = knn(sub_pos, pos, 1).

    pos [B, N, 3] float32 on the GPU.  `choices` (list of index tensors) overrides the random
    permutations (tests); otherwise torch.randperm(N, generator=generator)[:N // ratio].
    sample_method='fps' is the reference's other branch (:520-523): a farthest-point subset PER CLOUD ([B, S] picks,
    first pick = point 0) instead of one random subset shared by all clouds.

    sort='morton' (default when `choices` is None) first reorders every cloud along a Z-order curve
    (x, y, point_idx follow; the permutation is returned as ``data.order`` [B, N]) and keeps each random
    subset in ascending order, so that every level stays spatially sorted: neighbours then sit in
    nearby rows and the gather kernels hit L1/L2 instead of HBM.  The reference shuffles points
    anyway (semantic3d_dataset.py:435) and the network is permutation-equivariant, so results are
    the same cloud-by-cloud; kernels are correct for ANY order, this only buys locality."""
    if sort is None:
        sort = 'morton' if choices is None else 'none'
    if sort == 'morton':
        if order is None:                      # `order` [B, N]: a Morton permutation the caller already has (CollateGraph)
            order = morton_order(pos)

        movable = [t if (t is not None and torch.is_tensor(t) and t.dim() >= 2 and t.shape[1] == pos.shape[1]) else None
                   for t in (pos, x, y, point_idx)]
        moved = pick_rows(movable, order, per_cloud=True)
        pos, x, y, point_idx = [m if mv is not None else t for t, mv, m in zip((pos, x, y, point_idx), movable, moved)]
    elif sort != 'none':
        raise ValueError("sort must be 'morton' or 'none'")
    multiscale = []
    for i in range(num_scales):
        n = pos.shape[1]
        neighbor_idx = nearest_neighbors.knn_batch_device(pos, pos, kernel_size[i])
        method = sample_method.lower()
        if method not in ('random', 'fps'):
            raise NotImplementedError('Only `random` or `fps` sampling method is implemented!')      # the reference's message
        if choices is None and method == 'fps':
            choice = _fps_choice(pos, n // ratio[i])                               # [B, S], per cloud
            if sort == 'morton':
                choice = choice.sort(dim=1).values
            sub_pos, sub_idx = pick_rows([pos, neighbor_idx], choice, per_cloud=True)
            up_idx = nearest_neighbors.knn_batch_device(sub_pos, pos, 1)
            multiscale.append(Data(pos=pos, neighbor_idx=neighbor_idx, sub_idx=sub_idx, up_idx=up_idx))
            pos = sub_pos
            continue
        from .graph import to_device
        if choices is not None:
            choice = to_device(choices[i], pos.device)
        else:
            choice = torch.randperm(n, generator=generator)[: n // ratio[i]]
            if sort == 'morton':
                choice = choice.sort().values
            choice = to_device(choice.contiguous(), pos.device)        # (no stream synchronisation: graph.to_device)
        sub_pos, sub_idx = pick_rows([pos, neighbor_idx], choice, per_cloud=False)
        if kernel_size[i] >= 4:
            # the nearest subset member of a point is almost always one of its own K nearest neighbours: answered from the table
            # (bit-identical to the K = 1 search; `ranks`: the subsets' membership tables where the caller has them)
            up_idx = up_index_from_table(pos, neighbor_idx, choice, None if ranks is None else ranks[i], sub_pos=sub_pos)
        else:
            up_idx = nearest_neighbors.knn_batch_device(sub_pos, pos, 1)
        multiscale.append(Data(pos=pos, neighbor_idx=neighbor_idx, sub_idx=sub_idx, up_idx=up_idx))
        pos = sub_pos
    return MultiScaleData(x=x, y=y, point_idx=point_idx, cloud_idx=cloud_idx, multiscale=multiscale, order=order)


_PRIVATE_GEN = []


def _private_generator():
    if not _PRIVATE_GEN:
        _PRIVATE_GEN.append(torch.Generator().manual_seed(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF))
    return _PRIVATE_GEN[0]


class CollateGraph:
    """The per-batch preprocessing of a FIXED batch shape as ONE hipGraph replay: the device collate
    (``multiscale_compute``: Morton sort, kNN at every scale, subsets, up-indices) followed by ``target.load_`` (copies into
    the static batch a captured training step reads + in-place refresh of its neighbour tables, reverse CSRs and rel-pos
    moments).  Run eagerly the same work is ~470 launches and host-bound (5 ms of wall time for 3 ms of kernels).

    Nothing runs on the host per batch (device_draw=True, the default): the random subsets of
    datasets/semantic3d_dataset.py:517 are drawn INSIDE the graph by a counter-based kernel (crfconv_random_subsets: seeded
    by one draw on the caller's generator at construction -- ``state_dict()`` / ``load_state_dict()`` carry seed and counter
    for a resume -- and a device counter the graph advances, so every replay
    draws new subsets; the draws are NOT ``torch.randperm``'s) and the Morton argsort is this library's own scratch-free
    sort (crfconv_argsort_codes).  ``run`` = three copies into the static inputs + one replay.  device_draw=False keeps
    round 2's form: ``torch.randperm`` with the caller's generator on the host + a pinned upload, outside the graph.

        cg = CollateGraph(static_batch, generator=g)      # static_batch: the MultiScaleData the training graph was captured on
        cg.run(pos, x, y)                                 # new clouds [B, N, 3] / [B, N, C] / [B, N] on the device
        train_graph.replay()
    """

    def __init__(self, target, kernel_size=(16, 16, 16, 16, 16), ratio=(4, 4, 4, 4, 2), generator=None, device_draw=True, slot=0, gate=None):
        self.target, self.kernel_size, self.ratio, self.generator = target, tuple(kernel_size), tuple(ratio), generator
        self.device_draw = bool(device_draw)
        self.gate = gate                         # 4 int64 device words (crfconv_gate_wait): the graph's first launch waits for a mark of the training stream
        # The subset seed is a DRAW on the caller's generator (it advances the generator's state): graphs built one after the
        # other from one generator -- rebuilt per epoch, after a resume, the slots of a CollatePipeline -- get different
        # sequences, and a run re-started from the same generator state reproduces them.  (Keyed on initial_seed() alone, every
        # graph built from a generator replayed the SAME subset sequence.)  `slot` separates graphs built from equal states.
        if generator is None:
            # no caller's generator: ONE private generator per process, seeded once from the process seed and drawn from by every
            # graph built without a generator -- constructing a graph (a CollatePipeline builds one per slot) must not advance the
            # GLOBAL default generator (later initialisation / dropout / randperm draws would shift), and graphs built one after
            # the other (rebuilt per epoch, a second pipeline) must not replay the same subset sequence
            generator = _private_generator()
        self.generator = generator               # (device_draw=False draws its permutations from the same generator)
        drawn = int(torch.randint(0, 2 ** 62, (1,), generator=generator, dtype=torch.int64, device=generator.device).item())
        self.seed = (drawn + 0x632BE59BD9B4E019 * int(slot)) & 0xFFFFFFFFFFFFFFFF
        ms = target.multiscale
        dev = ms[0].pos.device
        self.pos = torch.empty_like(ms[0].pos)
        self.x = None if target.x is None else torch.empty_like(target.x)
        self.y = None if target.y is None else torch.empty_like(target.y)
        self.sizes = [lvl.pos.shape[1] for lvl in ms]
        self.order = torch.empty(ms[0].pos.shape[:2], dtype=torch.int64, device=dev)
        self.choices = [torch.empty(n // r, dtype=torch.int64, device=dev) for n, r in zip(self.sizes, self.ratio)]
        self.ranks = [torch.empty(n, dtype=torch.int32, device=dev) for n in self.sizes]      # membership tables of the subsets (device draw)
        self._pinned = [torch.empty(c.shape, dtype=torch.int64).pin_memory() for c in self.choices]
        # host buffers allocated once: a process that holds GPU memory pays for every large host malloc/free pair (the
        # unmap goes through the driver's MMU notifier -- measured 70-90 ms stalls in a loop that allocated per batch)
        self._perm = [torch.empty(n, dtype=torch.int64) for n in self.sizes]
        self._rank = [torch.empty(c.shape, dtype=torch.int64) for c in self.choices]
        self.graph = None
        self._uploaded = None                    # event after the last upload from the pinned buffers
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)      # batches collated so far (device_draw: keys the subsets)

    def state_dict(self):
        """What a checkpoint needs to continue this graph's subset sequence: the seed and the batch counter (device word)."""
        return {'seed': int(self.seed), 'counter': int(self.counter.item())}

    def load_state_dict(self, sd):
        self.seed = int(sd['seed']) & 0xFFFFFFFFFFFFFFFF
        self.counter.fill_(int(sd['counter']))
        if self.graph is not None and self.device_draw:
            self.graph = None                     # the seed is a launch scalar of the captured draw: capture again on the next run()

    def _draw(self):
        if self._uploaded is not None:
            self._uploaded.synchronize()         # the previous upload still owns the pinned buffers (long done in practice)
        for n, r, dst, pin, perm, rank in zip(self.sizes, self.ratio, self.choices, self._pinned, self._perm, self._rank):
            torch.randperm(n, generator=self.generator, out=perm)
            torch.sort(perm[: n // r], out=(pin, rank))                                          # ascending: levels stay sorted
            dst.copy_(pin, non_blocking=True)
        if self._uploaded is None:
            self._uploaded = torch.cuda.Event()
        self._uploaded.record()

    def _work_collate(self):
        if self.gate is not None:
            from . import _lib
            from .graph import ptr, stream_ptr
            _lib.call('crfconv_gate_wait', ptr(self.gate), CollatePipeline.GATE_MAX_WAIT_US, stream_ptr())
        if self.device_draw:
            from . import _lib
            from .graph import ptr, stream_ptr
            _lib.call('crfconv_add_i64', ptr(self.counter), 1, 1, stream_ptr())       # counter += 1 (a library launch: no framework kernel in the graph)
            random_subsets_device(self.sizes, [c.numel() for c in self.choices], self.seed, self.counter, self.choices, ranks=self.ranks)
            morton_order(self.pos, out=self.order)
        return multiscale_compute(self.pos, x=self.x, y=self.y, kernel_size=self.kernel_size, ratio=self.ratio,
                                  num_scales=len(self.sizes), choices=self.choices, sort='morton', order=self.order,
                                  ranks=self.ranks if self.device_draw else None)

    def _work(self):
        self.target.load_(self._work_collate())

    def _inputs(self, pos, x, y):
        self.pos.copy_(pos)
        if self.x is not None:
            self.x.copy_(x)
        if self.y is not None:
            self.y.copy_(y)
        if not self.device_draw:
            self._draw()                          # host torch.randperm + pinned upload; the argsort eagerly in front of the graph
            morton_order(self.pos, out=self.order)

    def run(self, pos, x=None, y=None):
        self._inputs(pos, x, y)
        if self.graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                count = self.counter.clone()
                self._work()                                  # warm-up outside the capture (allocator, lazy tables)
                self.counter.copy_(count)                     # ... which must not consume a batch number (resume: load_state_dict)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._work()
        self.graph.replay()
        return self.target

    # ---- the same work as TWO graphs, for a loop that trains on `target` itself (train.GraphedModel's static batch): the collate of the
    # NEXT batch (collate(): kNN etc. into this object's own staging tensors) may run on a side stream while the step of the current batch
    # still reads `target`; load() -- the copy into `target` and the in-place refresh of its tables, a fraction of the work -- then runs
    # on the training stream between two steps.  collate() of batch i + 2 must wait for load() of batch i + 1 (it overwrites the staging).
    def collate(self, pos, x=None, y=None):
        self._inputs(pos, x, y)
        if getattr(self, 'graph_collate', None) is None:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                count = self.counter.clone()
                self.target.load_(self._work_collate())       # warm-up of BOTH halves outside the captures
                self.counter.copy_(count)
            cur.wait_stream(side)
            torch.cuda.synchronize()
            self.graph_collate = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_collate):
                self._staged = self._work_collate()
            self.graph_load = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_load, pool=self.graph_collate.pool()):
                self.target.load_(self._staged)
        self.graph_collate.replay()

    def load(self):
        """target <- the batch collate() staged last (one replay on the current stream)."""
        self.graph_load.replay()
        return self.target


class CollatePipeline:
    """Per-batch preprocessing OVERLAPPED with training: two static batches, one ``CollateGraph`` each, the collate of
    batch i+1 on a side stream while the captured training step of batch i runs on the caller's stream.  Two hipGraphs on two
    streams do run concurrently on MI355X (graph BRANCHES do not): measured 6.4 ms per iteration for a 5.9 ms
    step + 2.2 ms collate graph at 4 x 40960 points.

        pipe = CollatePipeline([batch_a, batch_b], generator=g)       # the training step is captured once per batch
        pipe.submit(0, pos0, x0, y0)
        for i, (pos, x, y) in enumerate(next_clouds):
            s = i % 2
            batch = pipe.acquire(s)                # caller's stream waits for slot s's collate
            train_graph[s].replay()                # (launch first: the host part of the next collate hides behind it)
            pipe.submit(1 - s, pos, x, y, wait_current=False)   # side stream: waits only for slot 1 - s's release
            optimizer_graph.replay()
            pipe.release(s)                        # slot s may be overwritten once the work queued so far has run
    """

    GATE_MAX_WAIT_US = 3000       # a gated collate goes ahead after this long without a mark (crfconv_gate_wait)

    def __init__(self, batches, kernel_size=(16, 16, 16, 16, 16), ratio=(4, 4, 4, 4, 2), generator=None, device_draw=True,
                 priority=None, gate=False):
        """gate: the collate graphs START with a bounded device-side wait for a mark of the training stream (``self.mark()``,
        called inside the captured training step where its coarse levels begin -- e.g. PointConvBig.phase_hook): the side stream's
        kernels then fall into the part of the step whose launches leave most of the chip idle instead of beside its fine-level
        kernels (``mark_on('coarse_backward')``: 4.36 -> 4.25 ms per batch at 4 x 40 960 points; the forward's window or the collate as
        two gated graphs, one per window: 4.33 / 4.27).  Off until ``enable_gate(True)``; a collate that sees no mark within
        GATE_MAX_WAIT_US goes ahead, and after three such waits in a row the gate switches itself off (a marking stream that shares
        the side stream's hardware queue can never run beside the wait)."""
        import os
        import warnings
        if (torch.distributed.is_available() and torch.distributed.is_initialized()
                and int(os.environ.get('GPU_MAX_HW_QUEUES', '4')) < 8):
            # the mapping is fixed when the HIP runtime initialises (crfconv_amd/__init__.py picks it from RANK / WORLD_SIZE)
            warnings.warn('CollatePipeline under a process group with GPU_MAX_HW_QUEUES < 8: the collective\'s stream shares a '
                          'hardware queue with the collate stream (measured 7.1 instead of 5.8 ms per iteration); import '
                          'crfconv_amd before torch initialises the GPU, or export GPU_MAX_HW_QUEUES=8')
        self.batches = list(batches)
        self.gate = torch.zeros(4, dtype=torch.int64, device=self.batches[0].multiscale[0].pos.device) if gate else None
        self.graphs = [CollateGraph(b, kernel_size, ratio, generator, device_draw=device_draw, slot=k, gate=self.gate)
                       for k, b in enumerate(self.batches)]
        # the side stream at the LOWEST priority the device offers by default: the collate fills the CUs the training step
        # leaves idle (its many small launches) instead of taking turns with it
        if priority is None:
            priority = max(torch.cuda.Stream.priority_range())
        self.stream = torch.cuda.Stream(priority=priority)
        if self.gate is not None:
            # the gate needs a side stream that RUNS BESIDE the caller's: HIP maps streams onto a few hardware queues, and a wait on
            # the queue of the stream that is to mark it just times out (crfconv_gate_wait then switches the gate off).  Probe.
            for _ in range(8):
                if self.runs_beside_current(self.stream):
                    break
                self.stream = torch.cuda.Stream(priority=priority)
            else:
                warnings.warn('CollatePipeline(gate=True): no side stream found that runs beside the current one; the gate stays off')
                self.gate = None
                for g in self.graphs:
                    g.gate = None
        self._ready = [torch.cuda.Event() for _ in self.batches]
        self._free = [None for _ in self.batches]

    @staticmethod
    def runs_beside_current(stream, wait_us=5000):
        """True when a kernel on `stream` can wait for one launched later on the current stream (a gate handshake; a few milliseconds
        when it cannot)."""
        from . import _lib
        from .graph import ptr, stream_ptr
        probe = torch.zeros(4, dtype=torch.int64, device=torch.cuda.current_device())
        probe[2] = 1
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            _lib.call('crfconv_gate_wait', ptr(probe), int(wait_us), stream_ptr())
        _lib.call('crfconv_gate_mark', ptr(probe), stream_ptr())
        torch.cuda.synchronize()
        return int(probe[1].item()) == 1

    def mark(self, *_):
        """One mark on the gate from the CURRENT stream (a tiny launch): call it inside the captured training step."""
        if self.gate is not None:
            from . import _lib
            from .graph import ptr, stream_ptr
            _lib.call('crfconv_gate_mark', ptr(self.gate), stream_ptr())

    def mark_on(self, phase):
        """A phase hook (PointConvBig.phase_hook) that marks the gate when the model announces `phase`."""
        return lambda name, *_: self.mark() if name == phase else None

    def enable_gate(self, on=True):
        if self.gate is not None:
            self.gate[2] = 1 if on else 0
            self.gate[1] = self.gate[0]          # marks of the past open nothing

    def gate_timeouts(self):
        return 0 if self.gate is None else int(self.gate[3].item())

    def gate_is_on(self):
        """False once the gate has switched itself off (three waits in a row without a mark: see crfconv_gate_wait)."""
        return self.gate is not None and int(self.gate[2].item()) != 0

    def state_dict(self):
        """Seeds and batch counters of the slots' graphs (CollateGraph.state_dict): what a checkpoint needs to continue the subset sequences."""
        return {'graphs': [g.state_dict() for g in self.graphs]}

    def load_state_dict(self, sd):
        if len(sd['graphs']) != len(self.graphs):
            raise ValueError('CollatePipeline.load_state_dict: %d slots in the checkpoint, %d here' % (len(sd['graphs']), len(self.graphs)))
        for g, s in zip(self.graphs, sd['graphs']):
            g.load_state_dict(s)

    def submit(self, slot, pos, x=None, y=None, wait_current=True):
        """Queue the collate of `pos` / `x` / `y` into slot `slot` on the side stream.  wait_current=False: the inputs are
        known to be complete already (e.g. produced long ago, or on another stream the caller has synchronised with), so the
        side stream only waits for the slot's release -- call it AFTER launching the current training step and the host part
        of the collate (subset draw, argsort launches) hides behind that step too."""
        cg = self.graphs[slot]
        if cg.graph is None:                       # first use captures (with its own warm-up and a device synchronize)
            torch.cuda.synchronize()
            with torch.cuda.stream(self.stream):
                cg.run(pos, x, y)
                self._ready[slot].record()
            torch.cuda.synchronize()
            return
        if wait_current:
            self.stream.wait_stream(torch.cuda.current_stream())  # the caller produced pos / x / y on its stream
        if self._free[slot] is not None:
            self.stream.wait_event(self._free[slot])
        with torch.cuda.stream(self.stream):
            cg.run(pos, x, y)
            self._ready[slot].record()
        for t in (pos, x, y):                      # the caller may free these right away: the side stream still reads them
            if t is not None:
                t.record_stream(self.stream)

    def acquire(self, slot):
        torch.cuda.current_stream().wait_event(self._ready[slot])
        return self.batches[slot]

    def release(self, slot):
        if self._free[slot] is None:
            self._free[slot] = torch.cuda.Event()
        self._free[slot].record()
