"""Neighbour tables in the layout the kernels consume.

The reference hands the model int64 ``[B, n_tgt, K]`` per-cloud indices
(datasets/semantic3d_dataset.py:512-528).  A ``NeighborTable`` is the device-side plan derived
from one such tensor, built once per batch and shared by every layer that uses it:

  * ``idx32``   int32 ``[B * n_tgt, K]`` GLOBAL source rows (cloud b, id j -> b * n_src + j)
  * ``rev_ptr`` / ``rev_eid``  source-major CSR over the same edges (built lazily, on the first
    backward that needs it) so that every backward scatter is a deterministic gather.
"""
import torch

from . import _lib

_vp = _lib.ctypes.c_void_p


def ptr(t):
    """The device address of `t` for a C-ABI pointer argument (None = NULL).  A plain int: every entry point has its ctypes argtypes
    set (_lib.SIGNATURES), which take it as a pointer -- a c_void_p object per argument cost ~2 000 constructions per eager step."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream_ptr():
    """The current HIP stream of the current device as the C-ABI's crf_stream_t (the raw-handle query: a torch.cuda.Stream object per
    call cost ~10 us, a millisecond per eager training step)."""
    if _raw_stream is not None and _cur_device is not None:
        return _vp(_raw_stream(_cur_device()))
    return _vp(torch.cuda.current_stream().cuda_stream)


def to_device(t, device):
    """Host tensor -> device WITHOUT a stream synchronisation: through pinned memory with a non-blocking copy (torch's host allocator keeps
    the pinned block alive until the copy has run).  ``t.to(device)`` from pageable memory waits for the stream -- per call: five times per
    eager collate (the subset draws), twice per crop (jitter, shuffle) --, and on this stack a wait behind a short burst of work can
    return on a 10 ms tick.  Device tensors pass through."""
    if t.is_cuda:
        return t
    return t.pin_memory().to(device, non_blocking=True)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.CrfConvError(
                'crfconv_amd kernels run on MI355X only (got a %s tensor); there is no CPU path' % t.device)


class NeighborTable:
    __slots__ = ('idx32', 'idx16', 'B', 'n_tgt', 'n_src', 'K', '_rev', '_bad', '_checked', 'cache', 'n_edges', 'padded')

    def __init__(self, idx64, n_src, check=True):
        if idx64.dim() == 2:
            idx64 = idx64.unsqueeze(-1)
        if idx64.dim() != 3:
            raise ValueError('neighbour index must be [B, N, K], got %s' % (tuple(idx64.shape),))
        require_gpu(idx64)
        if idx64.dtype != torch.int64:
            idx64 = idx64.long()
        idx64 = idx64.contiguous()
        self.B, self.n_tgt, self.K = idx64.shape
        self.n_src = int(n_src)
        self.idx32 = torch.empty((self.B * self.n_tgt, self.K), dtype=torch.int32, device=idx64.device)
        self._bad = torch.zeros(1, dtype=torch.int32, device=idx64.device)
        # uint16 per-cloud ids as well when they fit: the streaming kernels then read half the index bytes
        self.idx16 = (torch.empty((self.B * self.n_tgt, self.K), dtype=torch.int16, device=idx64.device)
                      if self.n_src <= 65536 and self.K % 8 == 0 else None)
        # columns 1.. re-ordered by ascending source id (every consumer reduces over a row's columns; column 0,
        # which the CRF layer drops by position, stays): adjacent target rows then gather adjacent source rows
        _lib.call('crfconv_index_narrow_sorted', ptr(idx64), self.B, self.n_tgt, self.K, self.n_src, 1,
                  ptr(self.idx32), ptr(self.idx16), ptr(self._bad), stream_ptr())
        self._rev = None
        self._checked = False
        self.padded = False      # True: entries < 0 mean "no neighbour" (table_from_edges) -> generic kernels only
        self.n_edges = self.B * self.n_tgt * self.K
        self.cache = {}          # per-table memo (e.g. rel-pos moments shared by two ResNet blocks)
        if check == 'defer':
            self._validate_later()
        elif check:
            self.validate()

    def _validate_later(self):
        """The range check WITHOUT a host synchronisation: the bad-entry count travels to pinned memory behind the launch and
        ``check_pending()`` -- polled by the next table that is built or refreshed, callable by the user (``wait=True``) -- raises once
        it has arrived.  What ``table_of`` uses for the tables a model builds inside its forward: a synchronising ``.item()`` per table
        (thirteen per fresh batch) waits for everything queued before it and turns an asynchronous eager step into a synchronous one.
        Out-of-range entries were clamped by the narrowing kernel, so nothing can fault in the meantime; only the error comes later."""
        if torch.cuda.is_current_stream_capturing():
            return
        check_pending()
        host = _pinned_slot()
        host.copy_(self._bad, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _PENDING_CHECKS.append((ev, host, [self]))

    def refresh_(self, idx64, check=True):
        """New index values of the SAME shape into the SAME device buffers (narrowed table, and the reverse CSR and
        the memoised rel-pos moments if they exist): everything a captured hipGraph holds pointers to stays valid
        and now describes the new batch."""
        if idx64.dim() == 2:
            idx64 = idx64.unsqueeze(-1)
        if tuple(idx64.shape) != (self.B, self.n_tgt, self.K):
            raise ValueError('refresh_ needs the shape the table was built with %s, got %s'
                             % ((self.B, self.n_tgt, self.K), tuple(idx64.shape)))
        require_gpu(idx64)
        if idx64.dtype != torch.int64:
            idx64 = idx64.long()
        idx64 = idx64.contiguous()
        # (self._bad is cumulative: validate() resets it when it raises -- no one-word fill launch per table and refresh)
        if _BATCH['on']:                       # inside batched_reverse(): ONE narrowing launch for all tables at the exit
            _BATCH['narrow'].append((self, idx64, check))
            check = False                      # ... which also validates, once the counts exist
        else:
            _lib.call('crfconv_index_narrow_sorted', ptr(idx64), self.B, self.n_tgt, self.K, self.n_src, 1,
                      ptr(self.idx32), ptr(self.idx16), ptr(self._bad), stream_ptr())
        self._checked = False
        if self._rev is not None:
            self._build_reverse(*self._rev)
        for key, entry in list(self.cache.items()):
            if callable(getattr(entry, 'refresh_', None)):
                if _BATCH['on'] and callable(getattr(entry, 'batch_job', None)):
                    _BATCH['moments'].append((entry, self))      # after the narrowing, two launches for all tables
                else:
                    entry.refresh_(self)
        if check:
            self.validate()
        return self

    def validate(self):
        """One host sync: refuses tables with out-of-range entries (they were clamped, so nothing
        can fault, but the result would be meaningless)."""
        if not self._checked:
            if torch.cuda.is_current_stream_capturing():
                return             # no host sync inside a capture: out-of-range entries were clamped (nothing can fault),
                                   # the count stays in self._bad for a later validate()
            bad = int(self._bad.item())
            if bad:
                self._bad.zero_()
                raise IndexError('%d neighbour indices outside [0, %d)' % (bad, self.n_src))
            self._checked = True

    @property
    def m_tgt(self):
        return self.B * self.n_tgt

    @property
    def m_src(self):
        return self.B * self.n_src

    def _build_reverse(self, rev_ptr, rev_eid):
        if _BATCH['on']:                       # inside batched_reverse(): one set of launches for all tables at the end
            _BATCH['jobs'].append((self, rev_ptr, rev_eid))
            return
        E = self.m_tgt * self.K
        nbytes = _lib.load().crfconv_reverse_csr_workspace(E, self.m_src)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.idx32.device)
        _lib.call('crfconv_reverse_csr', ptr(self.idx32), E, self.m_src, ptr(rev_ptr), ptr(rev_eid),
                  ptr(ws), nbytes, stream_ptr())

    @property
    def reverse(self):
        if self._rev is None:
            dev = self.idx32.device
            rev_ptr = torch.empty(self.m_src + 1, dtype=torch.int32, device=dev)
            rev_eid = torch.empty(self.m_tgt * self.K, dtype=torch.int32, device=dev)
            self._build_reverse(rev_ptr, rev_eid)
            self._rev = (rev_ptr, rev_eid)
        return self._rev


_BATCH = {'on': False, 'jobs': [], 'narrow': [], 'moments': [], 'defer_check': False}
_PENDING_CHECKS = []          # (event, pinned counts, tables): validations whose device -> host copy is still in flight
_PINNED = {'ring': None, 'next': 0}


def _pinned_slot(dtype=torch.int32):
    """One int32 (or int64) of pinned host memory out of a ring of 1024 slots (allocated once: a pinned allocation per table would cost
    more than the synchronisation it replaces; a slot is reused long after its value has been read)."""
    if _PINNED['ring'] is None:
        _PINNED['ring'] = torch.zeros(1024, dtype=torch.int64).pin_memory()
    i = _PINNED['next']
    _PINNED['next'] = (i + 1) % 1024
    slot = _PINNED['ring'][i:i + 1]
    return slot if dtype == torch.int64 else slot.view(torch.int32)[:1]


def check_pending(wait=False):
    """Raises IndexError for tables refreshed under ``batched_reverse(defer_check=True)`` that held out-of-range entries (they
    were clamped: nothing faulted, but that batch's results were meaningless).  Looks only at copies that have arrived unless
    `wait`; no device synchronisation either way."""
    keep, bad = [], []
    for ev, host, tables in _PENDING_CHECKS:
        if not wait and not ev.query():
            keep.append((ev, host, tables))
            continue
        ev.synchronize()
        for n, t in zip(host.tolist(), tables):
            if n:
                t._bad.zero_()
                bad.append((n, t.n_src))
            else:
                t._checked = True
    _PENDING_CHECKS[:] = keep
    if bad:
        raise IndexError('; '.join('%d neighbour indices outside [0, %d)' % b for b in bad) + ' (in an earlier refreshed batch)')


class batched_reverse:
    """``with batched_reverse():`` around the refresh of several tables (MultiScaleData.load_): the per-table launches of
    NeighborTable.refresh_ are collected and issued on exit as batched calls with the same results -- ONE index-narrowing
    launch (crfconv_index_narrow_batched), the reverse CSRs by ONE crfconv_reverse_csr_batched call (five launches instead of
    seven per table), the rel-pos moments of the PointConv layers by ONE crfconv_pointconv_moments_batched call (two launches
    instead of two per table).  Nothing may read the refreshed tables before the exit.
    defer_check: the range check of the refreshed tables without a host synchronisation -- their bad-entry counts travel to pinned
    memory behind the launches and ``check_pending()`` (called by the next exit, or by the caller) raises once they have arrived."""

    def __init__(self, defer_check=False):
        self.defer_check = bool(defer_check)

    def __enter__(self):
        self.prev = _BATCH['on']
        self.prev_defer = _BATCH['defer_check']
        _BATCH['on'] = True
        _BATCH['defer_check'] = self.defer_check or self.prev_defer
        return self

    def __exit__(self, exc_type, *exc):
        _BATCH['on'] = self.prev
        defer, _BATCH['defer_check'] = _BATCH['defer_check'], self.prev_defer
        if self.prev:
            return False                       # nested: the outermost context flushes
        narrow, _BATCH['narrow'] = _BATCH['narrow'], []
        jobs, _BATCH['jobs'] = _BATCH['jobs'], []
        moments, _BATCH['moments'] = _BATCH['moments'], []
        if exc_type is not None:
            return False
        import ctypes
        lib = _lib.load()
        adr = lambda t: None if t is None else t.data_ptr()
        for i in range(0, len(narrow), 32):
            part = narrow[i:i + 32]
            arr = (_lib.NarrowJob * len(part))(*[_lib.NarrowJob(idx64.data_ptr(), t.B, t.n_tgt, t.K, t.n_src, 1, t.idx32.data_ptr(),
                                                                adr(t.idx16), t._bad.data_ptr()) for t, idx64, _ in part])
            _lib.call('crfconv_index_narrow_batched', ctypes.cast(arr, ctypes.c_void_p), len(part), stream_ptr())
        for i in range(0, len(jobs), 32):
            part = jobs[i:i + 32]
            arr = (_lib.RevJob * len(part))(*[_lib.RevJob(t.idx32.data_ptr(), t.m_tgt * t.K, t.m_src, rp.data_ptr(), re.data_ptr())
                                              for t, rp, re in part])
            p = ctypes.cast(arr, ctypes.c_void_p)
            nbytes = lib.crfconv_reverse_csr_batched_workspace(p, len(part))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=part[0][0].idx32.device)
            _lib.call('crfconv_reverse_csr_batched', p, len(part), ptr(ws), nbytes, stream_ptr())
        for i in range(0, len(moments), 16):
            part = moments[i:i + 16]
            arr = (_lib.MomentsJob * len(part))(*[entry.batch_job(t) for entry, t in part])
            p = ctypes.cast(arr, ctypes.c_void_p)
            nbytes = lib.crfconv_pointconv_moments_batched_workspace(p, len(part))
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=part[0][1].idx32.device)
            _lib.call('crfconv_pointconv_moments_batched', p, len(part), ptr(ws), nbytes, stream_ptr())
            for entry, _ in part:
                entry.mark_fresh()
        checks = [t for t, _, check in narrow if check]
        if defer and checks and not torch.cuda.is_current_stream_capturing():
            check_pending()                    # what earlier batches left behind and has arrived by now
            counts = torch.cat([t._bad for t in checks])
            host = torch.empty(len(checks), dtype=torch.int32, pin_memory=True)
            host.copy_(counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            _PENDING_CHECKS.append((ev, host, checks))
            return False
        if checks and not torch.cuda.is_current_stream_capturing():
            # ONE host synchronisation for all refreshed tables (round 5: one .item() per table -- thirteen queue drains per batch were
            # most of the 22 ms an eager MultiScaleData.load_ took)
            counts = torch.cat([t._bad for t in checks]).tolist()
            bad = []
            for n, t in zip(counts, checks):
                if n:
                    t._bad.zero_()
                    bad.append((n, t.n_src))
                else:
                    t._checked = True
            if bad:
                raise IndexError('; '.join('%d neighbour indices outside [0, %d)' % b for b in bad))
        return False


def table_from_edges(tgt, src, n_tgt, n_src, max_degree=64):
    """Variable-degree edge list -> padded NeighborTable.

    Edge e sends source row src[e] to target row tgt[e] (global row ids, int64 CUDA tensors).  Edges
    are grouped by target (stable: a target's edges keep their input order, which fixes the float
    summation order) into idx32 [n_tgt, Kp]; unused slots hold -1 ("no neighbour"), which every
    generic kernel skips.  Kp = the largest in-degree, at most `max_degree` (the reference's sparse
    layers bound it by construction: radius_graph(max_num_neighbors=k), knn_graph(k))."""
    require_gpu(tgt, src)
    E = tgt.numel()
    dev = tgt.device
    tab = NeighborTable.__new__(NeighborTable)
    tab.B, tab.n_tgt, tab.n_src = 1, int(n_tgt), int(n_src)
    tab._rev, tab._checked, tab.cache, tab.idx16 = None, True, {}, None
    tab.padded = True                     # -1 entries: the fast row kernels (no j < 0 test) must not see this table
    tab._bad = torch.zeros(1, dtype=torch.int32, device=dev)
    if E == 0:
        tab.K, tab.n_edges = 1, 0
        tab.idx32 = torch.full((n_tgt, 1), -1, dtype=torch.int32, device=dev)
        return tab
    if int(src.min()) < 0 or int(src.max()) >= n_src or int(tgt.min()) < 0 or int(tgt.max()) >= n_tgt:
        raise IndexError('edge endpoints outside [0, n)')
    order = torch.argsort(tgt, stable=True)
    ts, ss = tgt[order], src[order]
    deg = torch.bincount(ts, minlength=n_tgt)
    kmax = int(deg.max())
    if kmax > max_degree:
        raise _lib.CrfConvError('in-degree %d exceeds the supported maximum %d' % (kmax, max_degree))
    start = torch.cumsum(deg, 0) - deg
    rank = torch.arange(E, device=dev) - start[ts]
    idx = torch.full((n_tgt, kmax), -1, dtype=torch.int32, device=dev)
    idx[ts, rank] = ss.to(torch.int32)
    tab.K, tab.n_edges, tab.idx32 = kmax, E, idx
    return tab


def table_of(idx, n_src):
    """Table for an index tensor, memoised on the tensor object (a batch's neighbor_idx is used by
    two ResNet blocks and one CRF layer; sub_idx by the strided conv and its max-pool).  When the tensor was written
    in place since (a new batch copied into static buffers, ``_version`` changed) the existing table is REFRESHED in
    place -- same device buffers, so pointers recorded in a captured graph stay valid -- instead of replaced."""
    if isinstance(idx, NeighborTable):
        return idx
    cache = getattr(idx, '_crf_tables', None)
    if cache is None:
        cache = {}
        try:
            idx._crf_tables = cache
        except AttributeError:
            pass
    key = (int(n_src), idx.data_ptr(), tuple(idx.shape))
    hit = cache.get(key)
    if hit is None:
        tab = NeighborTable(idx, n_src, check='defer')       # (the range check without a host synchronisation: NeighborTable._validate_later)
        cache.clear()
        cache[key] = [tab, idx._version]
        return tab
    if hit[1] != idx._version:
        hit[0].refresh_(idx)
        hit[1] = idx._version
    return hit[0]
