"""The reference training step (trainval.py:99-106) as ONE hipGraph replay, for a caller that keeps its own loop.

    step = CapturedStep(model, optimizer, loss_fn, batch)      # batch: a MultiScaleData that stays resident (static buffers)
    for new_batch in loader:
        loss = step(new_batch)                                 # batch.load_(new_batch) + replay; `loss` is a device scalar

`optimizer` is any torch.optim optimizer whose step() is capturable as is (torch.optim.SGD with float hyper-parameters: a
learning-rate change needs a new CapturedStep) or optim.FlatSGD (then pass after_backward=bucket.pack).  `loss_fn(logits,
batch)` is the caller's, e.g. ``lambda o, d: F.cross_entropy(o, d.y.reshape(-1) - 1, weight=w, ignore_index=-1)``.
Drop (or detach) losses of earlier EAGER steps of the same model before constructing it: a live loss keeps that step's
autograd nodes -- the parameters' AccumulateGrad nodes, bound to the stream they were created on -- alive, and the capture
would then have to synchronise with that stream.

Round 6: the dense network captures ITSELF (``autograph``).  ``models.PointConvBig`` in training mode hands its forward to a private
GraphedModel (below) the first time it is called on a device batch, so the reference loop with nothing wrapped -- trainval.py:96-106 as
written -- runs as two hipGraph replays per step from its second iteration on.  What does not fit runs eagerly, silently and correctly:
eval mode, ``no_grad``, a batch of other shapes, a forward whose predecessor has not been through ``backward()`` yet (its output is still
in use: the replay would overwrite it), a model with forward / backward hooks on any submodule, a call during another stream capture.
``set_autograph(False)`` / ``CRFCONV_AUTOGRAPH=0`` / ``with no_autograph():`` switch it off (every launch issued by the host: 2-3 x the step
time); ``DistributedDataParallel`` around the model is untested -- switch it off there."""
import os

import torch

_AUTO = {'on': os.environ.get('CRFCONV_AUTOGRAPH', '1').strip().lower() not in ('0', 'false', 'off', 'no', ''), 'bypass': 0}


def set_autograph(on):
    """Process-wide switch of the self-capturing training forward (default on; CRFCONV_AUTOGRAPH=0 starts with it off)."""
    _AUTO['on'] = bool(on)


class no_autograph:
    """``with no_autograph():`` -- models called inside run eagerly (CapturedStep and GraphedModel use it around their own captures)."""

    def __enter__(self):
        _AUTO['bypass'] += 1
        return self

    def __exit__(self, *exc):
        _AUTO['bypass'] -= 1
        return False


def autograph_wanted(batch):
    if not _AUTO['on'] or _AUTO['bypass'] or not torch.is_grad_enabled():
        return False
    x = getattr(batch, 'x', None)
    if not (torch.is_tensor(x) and x.is_cuda and hasattr(batch, 'load_') and hasattr(batch, '_apply')):
        return False
    return not torch.cuda.is_current_stream_capturing()


def _has_hooks(model):
    for m in model.modules():
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, '_backward_pre_hooks', None):
            return True
    return False


def autograph_forward(model, batch):
    """The training-mode forward of a self-capturing model: the output of its private GraphedModel, or None -- run eagerly."""
    runner = model.__dict__.get('_autograph')
    if runner is None:
        if _has_hooks(model):
            return None
        runner = GraphedModel(model, guard_pending=True)
        object.__setattr__(model, '_autograph', runner)      # (not a submodule of the model: the model is the runner's)
    elif runner.training is not model.training:
        runner.training = model.training                     # (the runner is outside the model's tree: train() / eval() do not reach it)
    return runner(batch)


class CapturedStep:
    def __init__(self, model, optimizer, loss_fn, batch, after_backward=None, warmup=2, defer_weight_grads=True):
        """defer_weight_grads: run the backward inside ``ops.deferred_weight_grads()`` -- the ~150 weight-gradient launches of the
        pass (partial pass + sum per layer) go out as a dozen batched ones at its end; ``.grad`` of every parameter is set before
        ``optimizer.step()`` as usual (same partial slabs, fixed summation order).  False: the backward exactly as the caller wrote it."""
        self.model, self.optimizer, self.loss_fn, self.batch, self.after_backward = model, optimizer, loss_fn, batch, after_backward
        self.defer_weight_grads = bool(defer_weight_grads)
        with no_autograph():                                    # (the model's forward is issued launch by launch inside THIS capture)
            self._build(warmup)

    def _build(self, warmup):
        model, optimizer = self.model, self.optimizer
        # the warm-up steps (allocator, lazily built tables, momentum buffers) must not train: model state is put back afterwards,
        # momentum restarts from zero (mu * 0 + g = g: torch's first step, for dampening = 0)
        import copy
        supported = isinstance(optimizer, torch.optim.SGD) or type(optimizer).__name__ == 'FlatSGD'
        if supported and any(float(g.get('dampening', 0.0)) != 0.0 for g in optimizer.param_groups):
            supported = False                                   # (a zeroed momentum buffer gives (1 - d) g, not torch's first step g)
        if not supported and warmup > 0:
            raise TypeError('CapturedStep: the warm-up steps are undone for SGD-type optimizers without dampening only (torch.optim.SGD, '
                            'optim.FlatSGD); got %s -- pass warmup=0 and warm the caches up yourself' % type(optimizer).__name__)
        keep = [t.detach().clone() for t in list(model.parameters()) + list(model.buffers())]
        opt_state = copy.deepcopy(optimizer.state_dict())       # step counters, schedulers' view of the groups, any other state
        flat_steps = getattr(optimizer, 'steps', None)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
            with torch.no_grad():
                for t, k in zip(list(model.parameters()) + list(model.buffers()), keep):
                    t.copy_(k)
                had = bool(opt_state.get('state'))
                if had:
                    optimizer.load_state_dict(opt_state)          # an optimizer that had stepped before: exactly its old state
                for st in optimizer.state.values():
                    if not had and torch.is_tensor(st.get('momentum_buffer')):
                        st['momentum_buffer'].zero_()             # fresh optimizer: mu * 0 + g = g is torch's first step
                if hasattr(optimizer, 'buf'):
                    if 'flat_momentum' in opt_state:
                        optimizer.buf.copy_(opt_state['flat_momentum'])
                    else:
                        optimizer.buf.zero_()
                if flat_steps is not None:
                    optimizer.steps = flat_steps                  # FlatSGD's own counter (its check_every cadence)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.loss = self._step()

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(self.batch), self.batch)
        if self.defer_weight_grads:
            from . import ops
            with ops.deferred_weight_grads():
                loss.backward()
        else:
            loss.backward()
        if self.after_backward is not None:
            self.after_backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, new_batch=None):
        if new_batch is not None:
            self.batch.load_(new_batch)          # into the static buffers; tables, reverse CSRs, moments refreshed in place
        self.graph.replay()
        return self.loss


class _GraphedPass(torch.autograd.Function):
    """model(batch) as one forward replay; the gradient of whatever the caller computed from its output as one backward replay."""

    @staticmethod
    def forward(ctx, gm, *params):
        gm.fwd_graph.replay()
        ctx.gm = gm
        gm._pending = True                 # this output lives in the static buffer until its backward has run (autograph's guard)
        return gm.static_out.detach()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gm = ctx.gm
        gm._pending = False
        gm._keep_accumulated_grads()
        gm.static_gout.copy_(g)
        gm.bwd_graph.replay()
        # fresh tensor objects over the static buffers (as torch.cuda.make_graphed_callables returns them): autograd's AccumulateGrad
        # takes a gradient nobody else holds AS .grad -- handed the objects of gm.static_grads themselves it copies every one of them
        # (about 300 small copy launches per step for PointConvBig, 0.5 ms of the step)
        return (None,) + tuple(None if sg is None else sg.detach() for sg in gm.static_grads)


class GraphedModel(torch.nn.Module):
    """The reference loop UNCHANGED (trainval.py:99-106: zero_grad, model(data), the caller's loss, loss.backward(), optimizer.step())
    at replay cost: wrap the model once,

        net = GraphedModel(models.PointConvBig(6, 13, use_crf=True))       # the loop below it stays as written

    and every training-mode ``net(batch)`` is ONE hipGraph replay of the forward, ``loss.backward()`` ONE replay of the backward
    (the weight-gradient launches batched as in CapturedStep), with the loss and the optimizer the caller's own eager code in
    between -- about 300 library launches per step leave the host as two.  The first training call captures (after `warmup` eager
    passes on a side stream whose effect on BatchNorm statistics and dropout counters is undone); later batches of the SAME shapes
    are copied into the captured batch's buffers (MultiScaleData.load_: one copy launch + the table refreshes).  A batch of other
    shapes, eval mode and no_grad calls run the wrapped model eagerly.  Parameter gradients come back through autograd (hooks,
    accumulation over several backward passes see ordinary gradients -- tested; a hook-based wrapper such as DistributedDataParallel
    should too, untested); the output -- and, after ``backward()``, every parameter's
    ``.grad`` -- aliases a static buffer that the next call overwrites, as with torch.cuda.make_graphed_callables (gradients still held as ``.grad``
    at the next call are moved out first: _keep_accumulated_grads)."""

    def __init__(self, model, warmup=2, defer_weight_grads=True, guard_pending=False):
        """guard_pending (the self-capturing models' setting): a training call whose predecessor's output has not been through
        ``backward()`` runs the model eagerly instead of replaying over it."""
        super().__init__()
        self.model = model
        self.warmup, self.defer_weight_grads = int(warmup), bool(defer_weight_grads)
        self.guard_pending, self._pending, self._last_out = bool(guard_pending), False, None
        self.fwd_graph = self.bwd_graph = None
        self.static = self.static_out = self.static_gout = None
        self.params, self.static_grads, self._sig = [], [], None
        self._register_state_dict_hook(GraphedModel._strip_prefix_hook.__get__(self))
        self._register_load_state_dict_pre_hook(self._add_prefix_hook)

    # The wrapper is transparent for checkpoints: state_dict keys are the wrapped model's (reference checkpoints load with strict=True,
    # as on the bare model) -- through the module's OWN hooks (the 'model.' prefix is stripped on the way out and put back on the way
    # in), so that a GraphedModel that is a SUBMODULE of something else saves and loads symmetrically too (overriding state_dict /
    # load_state_dict covered the top-level call only: a parent's load recursed past the override and expected 'net.model.<key>').
    def _strip_prefix_hook(self, module, state_dict, prefix, local_metadata):
        inner = prefix + 'model.'
        for k in [k for k in state_dict if k.startswith(inner)]:
            state_dict[prefix + k[len(inner):]] = state_dict.pop(k)
        return state_dict

    def _add_prefix_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        inner = prefix + 'model.'
        for k in [k for k in state_dict if k.startswith(prefix) and not k.startswith(inner)]:
            state_dict[inner + k[len(prefix):]] = state_dict.pop(k)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__('model'), name)

    @staticmethod
    def _signature(batch):
        sig = []

        def visit(t):
            sig.append((tuple(t.shape), t.dtype))
            return t
        batch._apply(visit)
        return tuple(sig)

    def _backward(self):
        if self.defer_weight_grads:
            from . import ops
            with ops.deferred_weight_grads():
                self.static_out.backward(self.static_gout)
        else:
            self.static_out.backward(self.static_gout)

    def _capture(self, batch):
        with no_autograph():                                    # (the wrapped model's forward is issued launch by launch inside this capture)
            self._capture_eagerly_issued(batch)

    def _capture_eagerly_issued(self, batch):
        model = self.model
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._frozen = tuple(p.requires_grad for p in model.parameters())
        self.static = batch._apply(torch.clone)
        self._sig = self._signature(batch)
        user_grads = [p.grad for p in self.params]
        keep = [b.detach().clone() for b in model.buffers()]      # BatchNorm statistics, dropout counters: the warm-up must not count
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(self.warmup, 1)):                    # (at least one: lazily built tables, allocator, kernel modules)
                for p in self.params:
                    p.grad = None
                self.static_out = model(self.static)
                self.static_gout = torch.zeros_like(self.static_out)
                self._backward()
            with torch.no_grad():
                for b, k in zip(model.buffers(), keep):
                    b.copy_(k)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in self.params:
            p.grad = None
        self.fwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph, capture_error_mode='thread_local'):
            self.static_out = model(self.static)
        self.static_gout = torch.zeros_like(self.static_out)
        self.bwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool(), capture_error_mode='thread_local'):
            self._backward()
        self.static_grads = [p.grad for p in self.params]          # static buffers of the capture pool (None: not reached)
        # the captured autograd graph goes: it holds the parameters' AccumulateGrad nodes, created on the capture's stream, and a
        # node bound to another stream than its caller's costs a stream synchronisation per parameter in every later backward
        self.static_out = self.static_out.detach()
        for p, g in zip(self.params, user_grads):
            p.grad = g

    def _keep_accumulated_grads(self):
        """A caller who accumulates over several backward passes (no zero_grad() in between) holds last pass's static buffers as
        ``.grad`` (see _GraphedPass.backward): the next replay -- of the FORWARD already: the two graphs share one memory pool, a
        forward temporary may live where a gradient does -- is about to overwrite them, so they move to storage of their own first.
        The reference loop (zero_grad() every step) finds nothing to move."""
        for p, sg in zip(self.params, self.static_grads):
            if sg is not None and p.grad is not None and p.grad.data_ptr() == sg.data_ptr():
                p.grad = p.grad.clone()

    def _eager(self, batch):
        with no_autograph():
            return self.model(batch)

    def forward(self, batch):
        if not (self.training and torch.is_grad_enabled()):
            return self._eager(batch)
        if self.guard_pending and self._pending:
            if self._last_out is not None and self._last_out() is None:
                self._pending = False                           # ... unless nobody holds it any more (a step that skipped its backward)
            else:
                return self._eager(batch)                       # the previous output is still in use: nothing may be replayed over it
        if self.fwd_graph is None:
            self._capture(batch)
        elif batch is not self.static and self._signature(batch) != self._sig:
            return self._eager(batch)
        elif self.guard_pending and self._frozen != tuple(p.requires_grad for p in self.model.parameters()):
            return self._eager(batch)                           # parameters frozen / thawed since the capture
        self._keep_accumulated_grads()
        if batch is not self.static:
            self.static.load_(batch, defer_check=True)          # (no host synchronisation: a bad table raises at the next call)
        out = _GraphedPass.apply(self, *self.params)
        if self.guard_pending:
            import weakref
            self._last_out = weakref.ref(out)
        return out
