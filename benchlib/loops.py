"""The reference's training loop unchanged (trainval.py:96-106) on the HIP modules: eager, captured, graphed module, fresh batches."""
import time

import numpy as np
import torch

from .common import synth_cloud


def reference_loop(net, data, cw, steps):
    """The reference's training step, verbatim (trainval.py:99-106): optimizer.zero_grad(); y_pred = model(data);
    y = data.y.reshape(-1) - 1; loss = F.cross_entropy(y_pred, y, weight, ignore_index=-1); loss.backward(); optimizer.step()
    with torch.optim.SGD(lr=1e-2, momentum=0.95, weight_decay=1e-4) -- (a) eagerly, as a user who only swaps the import gets it,
    (b) through crfconv_amd.train.CapturedStep (the same five lines as one hipGraph replay)."""
    import torch.nn.functional as F
    from crfconv_amd.train import CapturedStep
    opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)

    def loss_fn(y_pred, d):
        return F.cross_entropy(y_pred, d.y.reshape(-1) - 1, weight=cw, ignore_index=-1)

    def one():
        opt.zero_grad()
        loss = loss_fn(net(data), data)
        loss.backward()
        opt.step()
        return loss.detach()      # (a live loss would keep this iteration's autograd nodes -- AccumulateGrad included -- alive)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e3
    out = {'eager_ms_per_step': eager, 'eager_final_loss': float(loss),
           'what': 'trainval.py:99-106 unchanged (zero_grad, model(data), F.cross_entropy(weight, ignore_index=-1), backward, '
                   'torch.optim.SGD.step), %d timed steps after 3 warm-up; eager_* : every launch issued by the host (train.set_autograph(False)); '
                   'bare_autograph_* : the same lines on the same bare model with the product default (the model captures its training forward / '
                   'backward itself, crfconv_amd.train)' % steps}
    # (a2) the SAME five lines on the SAME bare model, the product's default: the model hands its training forward to a private graph runner
    from crfconv_amd import train
    train.set_autograph(True)
    try:
        for _ in range(3):
            one()                              # (the first call captures)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = one()
        torch.cuda.synchronize()
        out['bare_autograph_ms_per_step'] = (time.perf_counter() - t0) / steps * 1e3
        out['bare_autograph_final_loss'] = float(loss)
        out['bare_autograph_replays'] = bool(getattr(net.__dict__.get('_autograph'), 'fwd_graph', None) is not None)
    finally:
        train.set_autograph(False)
        net.__dict__.pop('_autograph', None)
    for key, defer in (('captured_as_written_ms_per_step', False), ('captured_ms_per_step', True)):
        # CapturedStep's default batches the ~150 weight-gradient launches of the backward (ops.deferred_weight_grads inside the
        # capture; the caller's five lines are untouched); "as written" = the backward exactly as autograd issues it
        step = CapturedStep(net, opt, loss_fn, data, defer_weight_grads=defer)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        out[key] = (time.perf_counter() - t0) / steps * 1e3
        del step
    out['captured_final_loss'] = float(loss)
    # (c) the five lines verbatim again, on the model wrapped ONCE in crfconv_amd.train.GraphedModel: model(data) and loss.backward()
    # are one hipGraph replay each, F.cross_entropy and torch.optim.SGD.step stay the caller's eager code
    from crfconv_amd.train import GraphedModel
    bare, net = net, GraphedModel(net)
    opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    out['graphed_module_ms_per_step'] = (time.perf_counter() - t0) / steps * 1e3
    out['graphed_module_final_loss'] = float(loss)
    # (d) the loop the reference actually runs: `for data in train_loader:` (trainval.py:96) hands over a NEW collated batch every
    # step.  Here the collate is crfconv_amd.multiscale_compute on the device (the reference's runs in the DataLoader on the host,
    # datasets/semantic3d_dataset.py:501-534).  Four different raw batches take turns.
    import crfconv_amd
    from crfconv_amd.data import CollateGraph
    B, N = data.x.shape[:2]
    pool = []
    for r in range(4):
        clouds = [synth_cloud(9000 + 10 * r + i, N) for i in range(B)]
        pos_r = torch.from_numpy(np.stack([c[0] for c in clouds])).to(data.x.device)
        pool.append((pos_r, torch.cat([pos_r, torch.from_numpy(np.stack([c[1] for c in clouds])).to(pos_r.device)], -1),
                     torch.from_numpy(np.stack([c[2] for c in clouds])).to(pos_r.device)))
    gen = torch.Generator().manual_seed(4242)

    def one_on(d):
        opt.zero_grad()
        loss = loss_fn(net(d), d)
        loss.backward()
        opt.step()
        return loss.detach()

    def timed(body, n):
        for i in range(3):
            body(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            loss = body(3 + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, float(loss)
    # (d1) GraphedModel: the next batch's collate as a graph on a side stream while this step trains, its load into the captured batch as a
    # second graph between two steps (CollateGraph.collate / .load)
    cg = CollateGraph(net.static, generator=gen)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    staged, loaded = torch.cuda.Event(), torch.cuda.Event()
    cg.collate(*pool[0])                      # captures both graphs, stages batch 0
    staged.record()

    def graphed_fresh(i):
        main.wait_event(staged)
        cg.load()                             # captured batch <- staged batch (copy + in-place table refresh: one replay)
        loaded.record()
        side.wait_event(loaded)
        with torch.cuda.stream(side):
            cg.collate(*pool[(i + 1) % 4])    # kNN etc. of the NEXT batch beside this step
            staged.record()
        return one_on(net.static)
    out['fresh_graphed_ms_per_step'], out['fresh_graphed_final_loss'] = timed(graphed_fresh, steps)
    torch.cuda.synchronize()
    # (d2) the same, everything on one stream (collate graph, then the step)
    cg1 = CollateGraph(net.static, generator=gen)

    def graphed_fresh_serial(i):
        cg1.run(*pool[i % 4])
        return one_on(net.static)
    out['fresh_graphed_one_stream_ms_per_step'], _ = timed(graphed_fresh_serial, steps)
    net = bare
    opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)
    # (d3) nothing wrapped, nothing captured: eager collate + the five lines on the bare model
    def eager_fresh(i):
        pos_r, x_r, y_r = pool[i % 4]
        return one_on(crfconv_amd.multiscale_compute(pos_r, x=x_r, y=y_r, generator=gen, sort='morton'))
    out['fresh_eager_ms_per_step'], out['fresh_eager_final_loss'] = timed(eager_fresh, steps)
    # (d4) the same with the product default: eager collate, the bare model capturing itself (each new batch is loaded into the captured one)
    train.set_autograph(True)
    try:
        out['fresh_bare_autograph_ms_per_step'], _ = timed(eager_fresh, steps)
    finally:
        train.set_autograph(False)
        net.__dict__.pop('_autograph', None)
    out['fresh_what'] = ('a NEW batch every step (4 x 40 960-point clouds, device collate = the reference\'s _multiscale_compute_fn): fresh_eager = '
                         'crfconv_amd.multiscale_compute + the unchanged five lines on the bare model; fresh_graphed = train.GraphedModel + '
                         'data.CollateGraph.collate (side stream, beside the step) / .load (between steps); ..._one_stream = CollateGraph.run then the step')
    return out


