#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): M points/s, fwd+bwd(+SGD step) of PointConvBig with CRF decoders on
synthetic S3DIS-like clouds -- 4 clouds x 40960 points per GPU, K=16, 3 mean-field steps.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     the level-0 CRF mean-field forward (similarity + T steps) against the 8 TB/s HBM peak,
                  algorithmic bytes = m * (4 (K-1) + 4 H (2 T + 1))  (SURVEY.md 8(d)), timed with HIP events
  "cpu_baseline": the CPU oracle (oracle/crf_oracle.py, a port of the reference op sequence) on this host.
"""
import argparse
import json
import os
import sys
import time


def _self_launch():
    """`python bench.py --gpus N` WITHOUT a launcher's environment (no WORLD_SIZE): this process only starts the N ranks -- fresh
    children, one per GPU, with the torchrun variables (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT) --
    BEFORE it imports torch or crfconv_amd or touches the GPU (never a re-exec of a process that has), relays their output
    (rank 0 prints the JSON line), and exits non-zero if any rank fails.  N = 1 runs in-process as before unless --spawn asks
    for the one-rank process group (RCCL path with one rank).  Under torchrun (WORLD_SIZE set) nothing happens here."""
    import socket
    import subprocess
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--spawn', action='store_true')
    a, _ = ap.parse_known_args()
    if 'WORLD_SIZE' in os.environ or (a.gpus <= 1 and not a.spawn):
        return
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(max(1, a.gpus)):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(max(1, a.gpus)), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, t_fail = 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if bad and t_fail is None:
            rc, t_fail = bad[0], time.time()                     # a rank died: the others hang in their next collective
        if t_fail is not None and time.time() - t_fail > 20:
            for p in procs:
                if p.poll() is None:
                    p.kill()                                     # exactly the PIDs started above
    for p in procs:
        if p.returncode != 0 and rc == 0:
            rc = p.returncode
    sys.exit(rc if rc is not None else 1)


if __name__ == '__main__':
    _self_launch()

import numpy as np        # noqa: E402
import crfconv_amd        # noqa: E402,F401  (first: picks the hardware-queue mapping for a launch under a process group, crfconv_amd/__init__.py)
import crfconv_amd.train  # noqa: E402
crfconv_amd.train.set_autograph(False)      # this script captures its steps itself; reference_loop switches the models' own capture on for the legs that measure it
import torch              # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# everything but the driver lives in benchlib/ (the names stay importable from here: tests and scratch/ scripts use bench.<name>)
from benchlib.common import (BOX, HBM_PEAK, ROCPROF_MF, TRAFFIC_BWD, TRAFFIC_FWD, TRAFFIC_PC, TRAFFIC_STEP, VOX, _event_time,      # noqa: E402,F401
                             _meanfield_problem, _measured_traffic, _median_time, _rocprof_durations, copy_ceiling, cpu_share, make_batch, synth_cloud)
from benchlib.roofline import (_layer_summary, roofline_layer, roofline_meanfield, roofline_meanfield_bwd, roofline_pointconv,      # noqa: E402,F401
                               step_byte_model)
from benchlib.baselines import _parity_report, cpu_baseline      # noqa: E402,F401
from benchlib.configs import config_pipelines, other_configs      # noqa: E402,F401
from benchlib.loops import reference_loop      # noqa: E402,F401


def rehearse(args):
    """`bench.py --gpus N --rehearse`: the rank plumbing of the N-GPU run WITHOUT a GPU -- every rank joins a gloo group on the CPU
    (no HIP call: a box allows six processes on its card, an 8-GPU node is not ours to launch on), replicas are broadcast from rank
    0, K steps of [forward, backward, pack + failure flag, ONE flat all-reduce, update] run on a small CPU network through the SAME
    FlatGradAllReduce / broadcast_parameters / barrier + MAX-over-ranks timing code the real run uses, and rank 0 prints one JSON
    line.  What it catches before a multi-GPU lease is spent on it: port / environment propagation of the self-launch, a rank that
    never joins, unequal replicas, the straggler handling, thread oversubscription (N ranks x host threads).  It measures NOTHING
    about the kernels or RCCL: no scaling number exists until the driver's SCALE_r*.json has N > 1."""
    from crfconv_amd import distributed as D
    rank, world, local = D.init_from_env(use_gpu=False)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher environment says WORLD_SIZE=%d' % (args.gpus, world))
    torch.set_num_threads(1)
    torch.manual_seed(100 + rank)                            # different initial replicas: the broadcast must make them equal
    net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.LeakyReLU(0.1), torch.nn.Linear(32, 13))
    D.broadcast_parameters(net)
    bucket = D.FlatGradAllReduce(net)
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()
    g = torch.Generator().manual_seed(7 + rank)
    x, y = torch.randn(args.points, 6, generator=g), torch.randint(0, 13, (args.points,), generator=g)

    def step():
        bucket.zero()
        loss = torch.nn.functional.cross_entropy(net(x), y)
        loss.backward()
        bucket.pack()
        bucket.publish_guard()
        bucket.allreduce_packed()
        with torch.no_grad():
            for p, gview in zip(bucket.params, bucket.views):
                p.add_(gview, alpha=-1e-2 / world)
        return loss.detach()
    for _ in range(args.warmup):
        step()
    if grouped:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if grouped:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    spread = torch.zeros(1, dtype=torch.float64)
    if grouped:
        tt = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
        lo, hi = flat.clone(), flat.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        spread[0] = float((hi - lo).abs().max())
    if rank == 0:
        _print_json_line({'rehearsal': True, 'metric': 'rank plumbing only (CPU, gloo): NOT a benchmark', 'world': world, 'n_gpus': 0,
                          'dist_backend': torch.distributed.get_backend() if grouped else None, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': dt / max(args.steps, 1) * 1e3, 'replica_spread': float(spread[0]), 'guard_slot': float(bucket.guard),
                          'final_loss': float(loss), 'points_per_rank': args.points})
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


_JSON_FD = None


def _keep_stdout_for_the_json_line():
    """Under a process group the collective library writes a version banner to the process's stdout (RCCL 2.26: five lines per rank).  The
    contract is ONE JSON line on stdout: file descriptor 1 is pointed at stderr for the life of the process and the line goes out through a
    duplicate of the original descriptor."""
    global _JSON_FD
    if _JSON_FD is None and 'WORLD_SIZE' in os.environ:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def _print_json_line(obj):
    line = json.dumps(obj) + '\n'
    if _JSON_FD is None:
        sys.stdout.write(line)
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line.encode())


def main():
    _keep_stdout_for_the_json_line()
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--spawn', action='store_true', help='start the ranks as child processes even for --gpus 1 (one-rank RCCL group)')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4, help='clouds per GPU')
    ap.add_argument('--points', type=int, default=40960)
    ap.add_argument('--crf-steps', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pipe-gate', default='backward', choices=['off', 'forward', 'backward'],
                    help="the pipelined loop's device-side gate: the collate graph starts when the training step's forward ('forward') / backward ('backward') reaches its coarse levels, or wherever the launches fall ('off')")
    ap.add_argument('--mfma-min-rows', type=int, default=0, help='EXPERIMENT knob (A/B runs only): rows from which the row-streaming Linear forms take over from the tiled ones (ops.state.mfma_min_rows; 0 = the shipped 12288)')
    ap.add_argument('--rehearse', action='store_true', help='rank plumbing only, on the CPU over gloo (no GPU call, no benchmark): see rehearse()')
    ap.add_argument('--other-configs', action='store_true', help='(default since round 5; kept for old command lines)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the block that times BASELINE configs 3, 4 (per-GPU share) and 5 on this GPU')
    ap.add_argument('--graph', type=int, default=1, help='capture the training step into a hipGraph (1) or run eagerly (0)')
    ap.add_argument('--sort', default='morton', choices=['morton', 'none'],
                    help="point order emitted by the device collate (kernels are order-agnostic)")
    args = ap.parse_args()
    if args.rehearse:
        return rehearse(args)

    import crfconv_amd
    from crfconv_amd import distributed as D
    from crfconv_amd import models, ops

    if args.mfma_min_rows > 0:
        ops.state.mfma_min_rows = args.mfma_min_rows
    rank, world, local = D.init_from_env()
    torch.set_num_threads(max(1, min(torch.get_num_threads(), cpu_share() // max(1, world), 16)))     # 16 = the share per GPU of the pool's boxes
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher environment says WORLD_SIZE=%d' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    B, N, T, n_cls = args.batch, args.points, args.crf_steps, 13

    gen = torch.Generator().manual_seed(1234 + rank)
    make_batch(rank, B, N, dev, gen, args.sort)          # warm-up of the collate kernels
    data, t_pre = make_batch(rank, B, N, dev, gen, args.sort)
    torch.manual_seed(0)
    net = models.PointConvBig(6, n_cls, use_crf=True, steps=T).to(dev).train()
    D.broadcast_parameters(net)
    bucket = D.FlatGradAllReduce(net)
    # torch.optim.SGD(lr, momentum=0.95, weight_decay=1e-4) of trainval.py:69-72 as one launch over flat parameters
    opt = crfconv_amd.optim.FlatSGD(bucket, lr=1e-2, momentum=0.95, weight_decay=1e-4, grad_scale=1.0 / world)   # mean of the ranks' gradients inside the update
    cw = torch.ones(n_cls, device=dev)
    unit = torch.ones((), device=dev)                        # d loss / d loss, made once (loss.backward() fills a new one per call)

    # One training step = [A] zero grads, forward, weighted CE, backward, grads packed into ONE flat fp32 bucket
    #                     [C] all-reduce of that bucket over RCCL (only when world > 1; eager, never captured)
    #                     [B] SGD(momentum, weight decay) step on views of the bucket.
    # A and B are each captured into a hipGraph (the step issues >1000 small launches; replaying them removes
    # the Python host from the critical path).  The collective stays outside the graphs on purpose.
    def part_a(d=None):
        d = data if d is None else d
        opt.zero_grad()
        logits = net(d)
        loss = ops.training_loss(logits, d.y, cw, ignore_index=-1)             # trainval.py:101-104, fused kernel
        with ops.deferred_weight_grads(sink=bucket.view_of):     # one batched launch finishes all 74 dW / db reductions, into the bucket
            loss.backward(unit)
        bucket.pack()                                     # one batched copy into the flat bucket; .grad -> bucket views
        if world > 1:
            bucket.publish_guard()                        # this rank's barrier-failure flag into the slot the all-reduce sums with the gradients
        return loss.detach()

    def part_b():
        opt.step()                                           # reads bucket.flat, updates the flat parameter vector

    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()

    def collective():
        if grouped:
            bucket.allreduce_packed()                     # ONE all-reduce (sum) of gradients + failure flag; the 1 / world factor sits in the SGD kernel

    def step():
        loss = part_a()
        collective()
        part_b()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graph_note = 'eager'
    if args.graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, capture_error_mode='thread_local'):       # RCCL's watchdog thread keeps polling events
                static_loss = part_a()
            with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode='thread_local'):
                part_b()

            def step():                                   # noqa: F811
                ga.replay()
                collective()
                gb.replay()
                return static_loss
            graph_note = 'hipGraph replay (fwd+loss+bwd | eager RCCL all-reduce | SGD step)'
        except Exception as e:                            # capture not possible: stay eager, say so
            import traceback
            traceback.print_exc()
            torch.cuda.synchronize()
            graph_note = 'eager (graph capture failed: %s)' % str(e).splitlines()[0][:120]
    # ================================================================== THE TIMED REGION (the contract: W untimed warm-up steps, then
    # EXACTLY K steps between two barrier + torch.cuda.synchronize() pairs; MAX over ranks below).  Everything after it in this function
    # is diagnostics that never touches `value`.
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    # ================================================================== end of the timed region
    rank_ms = {'min': dt / args.steps * 1e3, 'max': dt / args.steps * 1e3}
    if grouped:                           # the contract's figure = MAX over ranks; min and max are both on the line so that a straggler shows
        tt = torch.tensor([dt, -dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt[0].item())
        rank_ms = {'min': -float(tt[1].item()) / args.steps * 1e3, 'max': dt / args.steps * 1e3}
    ms_per_step = dt / args.steps * 1e3
    value = world * B * N / (dt / args.steps) / 1e6

    # ---- per-batch costs that the timed loop (ONE resident batch, cached tables) does not pay, and the proof that the
    # captured step trains on a FRESH batch.  A new batch is collated (kNN etc. on the GPU), copied into the static
    # buffers the graph reads (MultiScaleData.load_: narrowed tables, reverse CSRs, rel-pos moments refreshed in place)
    # and the graph replayed; the same step is then repeated eagerly from the same weights on an independent collate of
    # the same clouds: loss and updated parameters must agree.
    def new_batch(seed0):
        clouds = [synth_cloud(seed0 + rank * B + i, N) for i in range(B)]
        pos_n = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
        x_n = torch.cat([pos_n, torch.from_numpy(np.stack([c[1] for c in clouds])).to(dev)], -1)
        y_n = torch.from_numpy(np.stack([c[2] for c in clouds])).to(dev)
        g2 = torch.Generator().manual_seed(seed0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = crfconv_amd.multiscale_compute(pos_n, x=x_n, y=y_n, generator=g2, sort=args.sort)
        torch.cuda.synchronize()
        return d, time.perf_counter() - t0

    # MultiScaleData.load_ of a fresh batch into the static buffers: device time by HIP events, host time of the call, and the wall time
    # per batch of a loop that loads batch after batch with ONE synchronisation at its end (what a training loop pays: it does not
    # synchronise per batch; load_(defer_check=True) makes no host synchronisation of its own).  A synchronising call right behind a
    # single load_ is reported too: on this stack a device synchronisation behind a short burst of work returns after 10 or 20 ms
    # (timer-tick granularity of the blocking wait) -- that, not the refresh, was round 5's "22 ms per fresh batch".
    fresh_pool, t_collate = [], []
    for rep in range(4):
        nd, tc = new_batch(5000 + 100 * rep)
        fresh_pool.append(nd)
        t_collate.append(tc)
    t_load_host, t_load_dev, t_load_sync = [], [], []
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        data.load_(fresh_pool[rep % 4], defer_check=True)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t_load_sync.append(time.perf_counter() - t0)
        t_load_host.append(t1 - t0)
        t_load_dev.append(e0.elapsed_time(e1) * 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(16):
        data.load_(fresh_pool[rep % 4], defer_check=True)
    torch.cuda.synchronize()
    t_load = [(time.perf_counter() - t0) / 16]
    # the same per-batch work as ONE hipGraph replay (data.CollateGraph: collate + in-place refresh of the static batch)
    from crfconv_amd.data import CollateGraph
    t_graph, cg_parts = None, None
    try:
        cg = CollateGraph(data, generator=torch.Generator().manual_seed(99 + rank))
        raw = [synth_cloud(7000 + rank * B + i, N) for i in range(B)]
        pos_r = torch.from_numpy(np.stack([c[0] for c in raw])).to(dev)
        x_r = torch.cat([pos_r, torch.from_numpy(np.stack([c[1] for c in raw])).to(dev)], -1)
        y_r = torch.from_numpy(np.stack([c[2] for c in raw])).to(dev)
        cg.run(pos_r, x_r, y_r)                           # captures
        torch.cuda.synchronize()
        ts = []
        for rep in range(7):
            t0 = time.perf_counter()
            cg.run(pos_r, x_r, y_r)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t_graph = float(np.median(ts))

        def part(fn):
            torch.cuda.synchronize()
            v = []
            for _ in range(5):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                v.append(time.perf_counter() - t0)
            return float(np.median(v)) * 1e3
        from crfconv_amd.data import morton_order
        cg_parts = {'host_work_ms': 0.0 if cg.device_draw else part(cg._draw),      # device_draw: subsets drawn inside the graph
                    'morton_argsort_alone_ms': part(lambda: morton_order(cg.pos, out=cg.order)),     # (inside the graph too)
                    'graph_replay_ms': part(cg.graph.replay), 'all_runs_ms': [round(t * 1e3, 2) for t in ts]}
    except Exception as e:                                # capture not possible: the eager figures above stand
        import traceback
        traceback.print_exc()
        torch.cuda.synchronize()
    nd, _ = new_batch(5200)
    data.load_(nd)                                        # the batch of seed 5200 for the comparison below
    torch.cuda.synchronize()
    buffers = {k: v.clone() for k, v in net.named_buffers()}
    flat0, mom0 = opt.flat.clone(), opt.buf.clone()
    loss_graph = float(step())                          # replay on the batch loaded last (seed 5200)
    torch.cuda.synchronize()
    flat_graph = opt.flat.clone()
    with torch.no_grad():
        opt.flat.copy_(flat0)
        opt.buf.copy_(mom0)
        for k, v in net.named_buffers():
            v.copy_(buffers[k])
    eager_data, _ = new_batch(5200)
    from crfconv_amd.graph import table_of
    tables_equal = True                                 # the refreshed static tables vs tables built from scratch
    for lvl_s, lvl_e in zip(data.multiscale, eager_data.multiscale):
        for name in ('neighbor_idx', 'sub_idx', 'up_idx'):
            a_, b_ = getattr(lvl_s, name, None), getattr(lvl_e, name, None)
            if a_ is None or not getattr(a_, '_crf_tables', None):
                continue
            for key, (tab_s, _) in a_._crf_tables.items():
                tab_e = table_of(b_, key[0])
                tables_equal &= bool(torch.equal(tab_s.idx32, tab_e.idx32))
                if tab_s._rev is not None:
                    tables_equal &= all(bool(torch.equal(u, v)) for u, v in zip(tab_s.reverse, tab_e.reverse))
    loss_eager = float(part_a(eager_data))
    collective()
    part_b()
    torch.cuda.synchronize()
    fresh = {'static_tables_equal_fresh_tables': tables_equal, 'loss_graph_replay': loss_graph,
             'loss_eager_same_weights': loss_eager,
             'max_param_diff_after_step': float((opt.flat - flat_graph).abs().max()),
             'collate_ms_per_batch': float(np.median(t_collate)) * 1e3,
             'table_refresh_ms_per_batch': float(np.median(t_load)) * 1e3, 'table_refresh_host_ms_per_batch': float(np.median(t_load_host)) * 1e3,
             'table_refresh_device_ms_per_batch': float(np.median(t_load_dev)) * 1e3,
             'table_refresh_plus_device_synchronize_ms': [round(v * 1e3, 2) for v in t_load_sync],
             'note': 'fresh batch -> MultiScaleData.load_ into the static buffers -> hipGraph replay, against an eager step '
                     'from the same weights and BatchNorm counters on an independent collate of the same clouds; the classifier\'s dropout mask '
                     'is keyed on (seed, step counter, element), so both runs draw the same mask and the losses agree; the refreshed '
                     'tables are compared bit for bit'}
    fresh['collate_plus_refresh_graph_ms_per_batch'] = None if t_graph is None else t_graph * 1e3
    fresh['collate_graph_parts'] = cg_parts if t_graph is not None else None
    per_batch_ms = ms_per_step + (t_graph * 1e3 if t_graph is not None else
                                  fresh['collate_ms_per_batch'] + fresh['table_refresh_ms_per_batch'])

    # ---- the training loop as it would run on a stream of fresh batches: two static batches, the collate graph of batch
    # i+1 on a side stream while the captured step of batch i trains (data.CollatePipeline).  EVERY iteration collates a
    # batch (host subset draw + Morton argsort + kNN at 5 scales + table / reverse-CSR / moment refresh) and trains on it.
    pipe_ms = None
    pipe_gate = None
    if t_graph is not None and graph_note.startswith('hipGraph'):
        try:
            from crfconv_amd.data import CollatePipeline
            data2, _ = make_batch(rank, B, N, dev, gen, args.sort)
            part_a(data2)                                 # builds the second batch's tables
            torch.cuda.synchronize()
            ga2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga2, pool=ga.pool(), capture_error_mode='thread_local'):
                part_a(data2)
            gas = (ga, ga2)
            gate = args.pipe_gate != 'off'
            raws = [(pos_r, x_r, y_r)]
            raw = [synth_cloud(9000 + rank * B + i, N) for i in range(B)]
            pos_q = torch.from_numpy(np.stack([c[0] for c in raw])).to(dev)
            raws.append((pos_q, torch.cat([pos_q, torch.from_numpy(np.stack([c[1] for c in raw])).to(dev)], -1),
                         torch.from_numpy(np.stack([c[2] for c in raw])).to(dev)))
            prio = os.environ.get('CRFCONV_BENCH_COLLATE_PRIORITY')     # A/B: side-stream priority (default: the lowest)
            pipe = CollatePipeline([data, data2], generator=torch.Generator().manual_seed(77 + rank),
                                   priority=None if prio is None else int(prio), gate=gate)
            pipe.submit(0, *raws[0])
            pipe.submit(1, *raws[1])                      # both collate graphs captured
            if gate:
                # the training graphs of the pipelined loop carry ONE more launch: a mark where the forward reaches its coarse levels
                # (PointConvBig.phase_hook); the collate graph on the side stream starts behind a bounded wait for it, so its kernels
                # fall beside the step's coarse-level launches (most of the chip idle) instead of beside its fine-level ones
                net.phase_hook = pipe.mark_on('coarse' if args.pipe_gate == 'forward' else 'coarse_backward')
                gated = []
                for d in (data, data2):
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg, pool=ga.pool(), capture_error_mode='thread_local'):
                        part_a(d)
                    gated.append(gg)
                net.phase_hook = None
                gas = tuple(gated)
                pipe.enable_gate(True)

            def run_pipe(n, i0):
                for i in range(i0, i0 + n):
                    slot = i % 2
                    pipe.acquire(slot)
                    gas[slot].replay()
                    # next batch: its host part (subset draw, argsort launches) now runs while the step above computes; the
                    # side stream waits only for the release of the slot it overwrites (the clouds were uploaded long ago)
                    pipe.submit(1 - slot, *raws[(i + 1) % 2], wait_current=False)
                    collective()
                    gb.replay()
                    pipe.release(slot)
            pipe.submit(0, *raws[0])
            sprio = os.environ.get('CRFCONV_BENCH_STEP_PRIORITY')       # A/B: the training graphs on a stream of this priority
            if sprio is not None:
                hp = torch.cuda.Stream(priority=int(sprio))
                hp.wait_stream(torch.cuda.current_stream())
                ctx = torch.cuda.stream(hp)
            else:
                import contextlib
                ctx = contextlib.nullcontext()
            with ctx:
                run_pipe(4, 0)
                barrier()
                t0 = time.perf_counter()
                run_pipe(args.steps, 4)
                barrier()
                pipe_ms = (time.perf_counter() - t0) / args.steps * 1e3
            if sprio is not None:
                torch.cuda.current_stream().wait_stream(hp)
            if grouped:
                tt = torch.tensor([pipe_ms], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                pipe_ms = float(tt.item())
            pipe_gate = {'mode': args.pipe_gate, 'timeouts': pipe.gate_timeouts(), 'still_on': pipe.gate_is_on()}
            pipe.enable_gate(False)
        except Exception:
            import traceback
            traceback.print_exc()
            torch.cuda.synchronize()
            pipe_ms = None

    # ---- the all-reduce alone (HIP events around 20 calls on the launching stream), and what the group looks like
    allreduce_us, allreduce_host_us, backend_name = None, None, None
    if grouped:
        backend_name = torch.distributed.get_backend()
        for _ in range(5):
            collective()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            collective()
        e1.record()
        torch.cuda.synchronize()
        allreduce_us = e0.elapsed_time(e1) / 20 * 1e3
        t0 = time.perf_counter()          # ... and what ISSUING it costs the host (the call sits between two graph replays)
        for _ in range(20):
            collective()
        allreduce_host_us = (time.perf_counter() - t0) / 20 * 1e6
        torch.cuda.synchronize()

    # ---- what a caller of the UNCHANGED reference loop gets (trainval.py:99-106), on the same model and batch: no graph, no
    # FlatSGD, no deferred weight gradients, no fused loss -- torch.optim.SGD and F.cross_entropy as the reference writes them;
    # and the same loop handed to crfconv_amd.train.CapturedStep (one hipGraph replay per step, still torch.optim.SGD).
    ref_loop = None
    if rank == 0:
        try:
            ref_loop = reference_loop(net, data, cw, args.steps)
        except Exception as e:
            import traceback
            traceback.print_exc()
            torch.cuda.synchronize()
            ref_loop = {'error': str(e).splitlines()[0][:200]}

    if rank == 0:
        out = {
            'metric': 'M points/sec fwd+bwd, S3DIS 40960-pt cloud, K=16, 3 CRF iters; mIoU parity',
            'value': value, 'unit': 'M points/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: S3DIS-like synthetic clouds (one point per 4 cm voxel of an '
                                   '8x8x3 m box), %d clouds x %d pts per GPU, K=16, ratios [4,4,4,4,2], '
                                   'PointConvBig(in=6, classes=13, use_crf, steps=%d), train mode: fwd + weighted CE + '
                                   'bwd + flat-bucket grad all-reduce + SGD(momentum) step; tables resident in HBM; point order from '
                                   'the device collate: %s' % (B, N, T, args.sort),
                       'global_batch': world * B, 'points_per_cloud': N, 'parallelism': 'dp%d (batch-sharded)' % world},
            'final_loss': float(loss),
            'preprocess_ms_per_batch': fresh['collate_ms_per_batch'],
            'table_refresh_ms_per_batch': fresh['table_refresh_ms_per_batch'],
            'preprocess_plus_refresh_graph_ms_per_batch': fresh['collate_plus_refresh_graph_ms_per_batch'],
            'value_incl_preprocessing_serial': world * B * N / (per_batch_ms * 1e-3) / 1e6,
            'pipelined_ms_per_batch': pipe_ms, 'pipelined_gate': pipe_gate,
            'value_incl_preprocessing': world * B * N / ((pipe_ms or per_batch_ms) * 1e-3) / 1e6,
            'fresh_batch_replay': fresh,
            'launch_mode': graph_note,
            'rccl_ranks_seen': torch.distributed.get_world_size() if grouped else 1,
            'dist_backend': backend_name, 'allreduce_us': allreduce_us, 'allreduce_host_issue_us': allreduce_host_us,
            'rank_ms_per_step': rank_ms, 'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
            'trainval_eager_ms_per_step': None if ref_loop is None else ref_loop.get('eager_ms_per_step'),
            'trainval_captured_ms_per_step': None if ref_loop is None else ref_loop.get('captured_ms_per_step'),
            'trainval_graphed_module_ms_per_step': None if ref_loop is None else ref_loop.get('graphed_module_ms_per_step'),
            'trainval_captured_as_written_ms_per_step': None if ref_loop is None else ref_loop.get('captured_as_written_ms_per_step'),
            'reference_loop': ref_loop,
            'parity': _parity_report(),
        }
        out['roofline'] = roofline_meanfield(data, dev, 8, T)
        out['roofline']['measured_copy_GBps'] = copy_ceiling(dev)
        out['roofline_bwd'] = roofline_meanfield_bwd(data, dev, 8, T)
        if out['roofline_bwd'] is not None:
            # forward and backward of the level-0 layer as ONE figure: algorithmic bytes of both over the sum of their
            # average times (each timed on its own, back to back on the launching stream)
            rf, rb = out['roofline'], out['roofline_bwd']
            alg = rf['alg_bytes_per_launch'] + rb['alg_bytes_per_launch']
            t = (rf['avg_launch_us'] + rb['avg_launch_us']) * 1e-6
            tr = None if rf['traffic'] is None or rb['traffic'] is None else rf['traffic'] + rb['traffic']
            out['roofline_fwd_bwd'] = {'bound': 'hbm', 'achieved': alg / t / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                                       'frac': alg / t / HBM_PEAK, 'traffic': tr, 'alg_bytes_per_launch': alg,
                                       'avg_launch_us': t * 1e6,
                                       'kernel': 'level-0 mean-field forward + backward (roofline + roofline_bwd)'}
        try:
            out['roofline_layer'] = roofline_layer(data, dev, T, level0=(out['roofline'], out['roofline_bwd']) if out['roofline_bwd'] else None)
        except Exception as e:                             # a measurement beside the contract line, never fatal
            out['roofline_layer'] = {'error': str(e).splitlines()[0][:200]}
        try:
            out['roofline_pointconv'] = roofline_pointconv(data, dev, 8)
        except Exception as e:                             # a measurement beside the contract line, never fatal
            out['roofline_pointconv'] = {'error': str(e).splitlines()[0][:200]}
        # the whole step against the HBM roofline: algorithmic bytes (step_byte_model + 20 B per parameter for SGD with momentum) beside
        # the HBM-side traffic measured by rocprofv3 PMC passes of this build (profiles/, null when the sources changed since)
        n_par = sum(p.numel() for p in net.parameters())
        alg, groups = step_byte_model(B, N, 16, T, n_cls)
        alg += 20 * n_par
        step_traffic, step_note = _measured_traffic(TRAFFIC_STEP, {'B': B, 'N': N, 'K': 16, 'T': T})
        out['roofline_step'] = {'bound': 'hbm', 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'alg_bytes_per_step': alg,
                                'achieved': alg / (ms_per_step * 1e-3) / 1e9, 'frac': alg / (ms_per_step * 1e-3) / HBM_PEAK,
                                'traffic': step_traffic, 'traffic_source': step_note,
                                'waste_ratio': None if step_traffic is None else step_traffic / alg,
                                'frac_on_measured_traffic': None if step_traffic is None else step_traffic / (ms_per_step * 1e-3) / HBM_PEAK,
                                'groups': groups, 'ms_per_step': ms_per_step,
                                'kernel': 'the whole captured training step (fwd + weighted CE + bwd + SGD), %d parameters' % n_par}
        if world == 1 and not args.no_other_configs:          # every BASELINE config in the line: a few seconds on one GPU
            try:
                out['other_configs'] = other_configs(dev)
            except Exception as e:
                import traceback
                traceback.print_exc()
                torch.cuda.synchronize()
                out['other_configs'] = {'error': str(e).splitlines()[0][:200]}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(data, net, T, data.y, n_cls, dev)
        _print_json_line(out)
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
