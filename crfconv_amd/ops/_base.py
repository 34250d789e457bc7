"""Shared pieces of the operator modules: conversions, per-stream zero words, grid-barrier workspace and its failure check, run-time switches."""
import ctypes

import torch

from .. import _lib
from ..graph import NeighborTable, ptr, require_gpu, stream_ptr

BN_EPS = 1e-5


def _f32c(t):
    if t.dtype is torch.float32 and t.is_contiguous():
        return t.detach()
    return t.detach().to(torch.float32).contiguous()


def _pad_channels(t, C):
    """Zero-pad the last dim to C (kernels take power-of-two channel counts; zeros are exact)."""
    if t.shape[-1] == C:
        return t
    return torch.nn.functional.pad(t, (0, C - t.shape[-1]))


def _next_supported(c, choices):
    for v in choices:
        if c <= v:
            return v
    raise _lib.CrfConvError('channel count %d exceeds the largest supported (%d)' % (c, choices[-1]))


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])



_TICKETS = {}
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_key(device):
    """(device index, handle of the stream the caller launches on): the inter-workgroup scratch words below are per stream,
    so that launches on two streams of one device (a second training stream, evaluation beside training) never share
    barrier / ticket counts.  CAPTURED launches of a device all use ONE buffer (created by the eager warm-up pass, whatever
    stream the capture later runs on): hipGraphs that contain mean-field backward or one-launch MLP kernels must therefore be
    replayed one after the other (the loops of this package do) -- replaying two of them CONCURRENTLY on different streams is
    unsupported, their barrier and ticket counts would mix."""
    idx = device.index if isinstance(device, torch.device) else None      # (what this package passes: no torch.device() round trip)
    if idx is None:
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if torch.cuda.is_current_stream_capturing():
        return idx, 'capture'                      # captured launches: one buffer per device, created by the eager warm-up pass
    if _raw_stream is not None:                    # the raw-handle query (a torch.cuda.Stream object per call costs ~3 us, ~100 calls a step)
        return idx, int(_raw_stream(idx))
    return idx, int(torch.cuda.current_stream(idx).cuda_stream)


def _stream_buf(table, device, make):
    """table[(device, stream)] (see _stream_key), created on first use; the eager pass that creates a stream's buffer also
    creates the device's capture buffer, so that a capture never allocates."""
    key = _stream_key(device)
    buf = table.get(key)
    if buf is None:
        buf = table[key] = make()
        if key[1] != 'capture' and (key[0], 'capture') not in table:
            table[(key[0], 'capture')] = make()
    return buf


def _ticket(device):
    """Zero words for the "last workgroup finishes" reductions (left zero by the kernels), per (device, stream)."""
    return _stream_buf(_TICKETS, device, lambda: torch.zeros(_lib.load().crfconv_ticket_bytes() // 4, dtype=torch.int32, device=device))




def _pc_ticket(device):
    return ptr(_ticket(device))




def _mlp_ticket(device):
    return ptr(_ticket(device))


_sync_ws = {}


def gridsync_ws(dev):
    """The barrier words of the one-launch kernels (csrc/gridsync.hpp): zeroed once, left zero by every launch.  One buffer
    per (device, stream): launches that share one are ordered by their stream (see _stream_key for captured graphs)."""
    return _stream_buf(_sync_ws, dev, lambda: torch.zeros(_lib.load().crfconv_gridsync_workspace() // 4, dtype=torch.int32, device=dev))




def fail_word_ptrs(dev):
    """Device addresses of the sticky barrier-failure words of the barrier workspaces of `dev`, at most 8 (the kernel-side guard's
    capacity): ALWAYS the buffer captured graphs use and the current stream's, then the other streams' newest first.  A process that has
    used more than that many eager streams gets a warning -- the guard then does not see the oldest streams' words (check_gridsync
    still does: it reads every workspace)."""
    want = torch.device(dev).index
    if want is None:
        want = torch.cuda.current_device()
    word = _lib.load().crfconv_gridsync_fail_word()
    keys = [k for k in _sync_ws if k[0] == want]            # dict order = creation order
    first = [k for k in ((want, 'capture'), (want, int(torch.cuda.current_stream(want).cuda_stream))) if k in _sync_ws]
    rest = [k for k in reversed(keys) if k not in first]
    if len(first) + len(rest) > 8:
        import warnings
        warnings.warn('fail_word_ptrs: %d barrier workspaces on device %d, the update guard watches 8 (capture buffer, current stream, '
                      'newest streams); ops.check_gridsync() reads all of them' % (len(first) + len(rest), want))
    return [_sync_ws[k].data_ptr() + 4 * word for k in (first + rest)[:8]]


def check_gridsync(dev=None, reduced_flag=None):
    """Raises CrfConvError when a one-launch kernel's grid barrier has timed out on `dev` since the last check (its
    workgroups were not all resident -- CU mask, reserved CUs; the launch's outputs were NaN-poisoned).  One 4-byte
    device read per barrier workspace (a synchronisation): FlatSGD.step() calls it every `check_every` eager steps, a loop that
    replays captured graphs should call it once per epoch / logging interval.  `reduced_flag`: the guard slot of the gradient bucket
    (distributed.FlatGradAllReduce.guard) -- under data parallelism it holds the SUM of the ranks' flags after the all-reduce, so
    every rank raises in the same step, not only the one whose kernel failed (the others would hang in the next collective).
    What survives a failure: PARAMETERS and the MOMENTUM buffer on every rank -- FlatSGD's update kernel reads the same sticky
    words and the reduced slot and changes nothing while one is set (eager steps and captured replays alike), so every step
    since the failure was a no-op for them.  What does not: the BatchNorm RUNNING statistics of the layers downstream of the failed
    launch saw NaN activations in those steps (the failed layer itself skips its update) -- restore the model's buffers from the
    last checkpoint, or reset them, before going on.  After a failure the one-launch KERNEL is switched off for the rest of the
    process (the small-MLP nodes go on with a launch-separated forward: tiled product + BatchNorm launches, _small_fwd), so a
    caller that catches the error and has repaired the buffers can re-run the step."""
    word = _lib.load().crfconv_gridsync_fail_word()
    bad = []
    want = None if dev is None else torch.device(dev).index
    for table in (_sync_ws,):
        for key, ws in list(table.items()):
            if want is not None and key[0] is not None and key[0] != want:
                continue
            code = int(ws[word].item())
            if code != 0:
                ws.zero_()                                    # barrier counts and the flag: a clean slate for the retry
                bad.append((key, code))
    remote = False
    if reduced_flag is not None and reduced_flag.numel():
        v = float(reduced_flag.reshape(-1)[0].item())
        remote = not (v == 0.0)
        if remote:
            reduced_flag.zero_()
    if bad or remote:
        state.small_mlp_disabled = True
        where = ('device(s) %s (code 0x%x)' % (sorted({k[0] for k, _ in bad}), bad[0][1])) if bad else 'another rank of the process group'
        raise _lib.CrfConvError('grid barrier timed out on %s: a one-launch kernel could not get all its '
                                'workgroups resident; its outputs were poisoned with NaN.  The one-launch MLP path is now '
                                'disabled for this process (CRFCONV_NO_ONE_LAUNCH_MLP=1 does the same up front).' % where)




class _State:
    """Switches the tests and bench.py's experiment knobs flip at run time (one object: the operator modules read it, a caller sets
    ``ops.state.<name>``)."""
    no_join = False              # tests: True = lin_out, bn_apply and add_lrelu as separate nodes (the fused nodes must give the same results)
    no_fork = False              # tests: True = autograd's own accumulation pass instead of the fork chain
    no_prefold = False           # tests: True = one fold launch inside every PointConv layer
    # set by check_gridsync after a barrier failure (CRFCONV_NO_ONE_LAUNCH_MLP: from the start): launch-separated forward from then on
    small_mlp_disabled = __import__('os').environ.get('CRFCONV_NO_ONE_LAUNCH_MLP') is not None
    # the two launches of a coarse-level MLP backward (tile sums, dX product) as one whose product workgroups wait for the sums inside
    # the launch (csrc/gemm.hip mlp_small_bwd_jobs_kernel); CRFCONV_SMALL_BWD_TWO_LAUNCHES=1: A/B runs of bench.py
    small_bwd_one_launch = __import__('os').environ.get('CRFCONV_SMALL_BWD_TWO_LAUNCHES') is None
    # the backward of the CRF layers' matrices as riders of the MLP blocks' end-of-pass weight-gradient launch (defer._flush_mlp_dw)
    dw_hosts_mats = __import__('os').environ.get('CRFCONV_NO_TAIL_RIDERS') is None
    # below this many rows the tiled product (gemm.hip) and the small-MLP nodes; swept on the step: 4096 -> 4.733 ms, 12288 (the
    # 10 240-row level joins the small forms) -> 4.694 ms, 65536 -> 4.736 ms
    mfma_min_rows = 12288
    # the mean-field forward as one launch with block-resident rows (csrc/crf_block.hip; ops.crf._block_rows): 'auto' | 'on' | 'off'
    mf_block = __import__('os').environ.get('CRFCONV_MF_BLOCK', 'auto')      # (the environment variable: A/B runs of bench.py)
    mf_block_min_rows = 65536        # below: too few workgroups of 256+ rows to fill the chip
    mf_block_min_locality = 0.6      # fraction of table entries inside the target's own block (Morton-sorted clouds: ~0.8)


state = _State()
