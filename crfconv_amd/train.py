"""The reference training step (trainval.py:99-106) as ONE hipGraph replay, for a caller that keeps its own loop.

    step = CapturedStep(model, optimizer, loss_fn, batch)      # batch: a MultiScaleData that stays resident (static buffers)
    for new_batch in loader:
        loss = step(new_batch)                                 # batch.load_(new_batch) + replay; `loss` is a device scalar

`optimizer` is any torch.optim optimizer whose step() is capturable as is (torch.optim.SGD with float hyper-parameters: a
learning-rate change needs a new CapturedStep) or optim.FlatSGD (then pass after_backward=bucket.pack).  `loss_fn(logits,
batch)` is the caller's, e.g. ``lambda o, d: F.cross_entropy(o, d.y.reshape(-1) - 1, weight=w, ignore_index=-1)``.
Drop (or detach) losses of earlier EAGER steps of the same model before constructing it: a live loss keeps that step's
autograd nodes -- the parameters' AccumulateGrad nodes, bound to the stream they were created on -- alive, and the capture
would then have to synchronise with that stream."""
import torch


class CapturedStep:
    def __init__(self, model, optimizer, loss_fn, batch, after_backward=None, warmup=2, defer_weight_grads=True):
        """defer_weight_grads: run the backward inside ``ops.deferred_weight_grads()`` -- the ~150 weight-gradient launches of the
        pass (partial pass + sum per layer) go out as a dozen batched ones at its end; ``.grad`` of every parameter is set before
        ``optimizer.step()`` as usual (same partial slabs, fixed summation order).  False: the backward exactly as the caller wrote it."""
        self.model, self.optimizer, self.loss_fn, self.batch, self.after_backward = model, optimizer, loss_fn, batch, after_backward
        self.defer_weight_grads = bool(defer_weight_grads)
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
