#!/usr/bin/env python3
"""bench.py -- forward+backward Mpixels/s of the dirt rasterise hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches it under torch.distributed.run, one rank per GPU.  A "step" is one
dirt_rasterise_forward + one dirt_rasterise_backward over one batch of synthetic scenes resident in
HBM.  Rank 0 prints ONE JSON line.

Workload: BASELINE.json's metric config K3 -- a 10 000-triangle random mesh rendered at
1024x1024x4 channels (SURVEY.md 8d `rand_mesh`, seed 0 + scene index), `--scenes-per-gpu` scenes per
rank: 1 by default AT EVERY N (K3 itself on each GPU), so that value(N) / value(1) is weak scaling of one
and the same per-GPU workload; `--scenes-per-gpu 8 --gpus 8` is K4 (64 scenes over 8 GPUs) and
`--scenes-per-gpu 64 --gpus 1` K4 on one GPU.  Scenes are independent, so ranks share nothing on the data
path (no collective inside the timed region; DESIGN.md "Multi-GPU").  At N > 1 rank 0 also times the
same per-GPU workload ALONE (the other ranks idle at a barrier) and prints it as `scaling_reference`;
`--gather` times, separately from the render, the collection of all ranks' pixels on rank 0 over RCCL
("report the gathered variant separately").

Extra objects on the JSON line:
  roofline     -- the dominant kernel (by summed HIP-event time over the timed steps) against the HBM
                  roofline: achieved = that kernel's algorithmic bytes per launch / its mean duration.
  cpu_baseline -- the CPU oracle ("port" of the reference's algorithm; the reference op itself is
                  GPU-only) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# RCCL / cross-process sharing of device memory needs dmabuf IPC on this pool's host driver (without it
# hipIpcGetMemHandle fails with "invalid argument"); exported by the environment already -- kept here for any other launcher
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec ...
HBM_COPY_GBPS = 6290.0  # ... and 6.29 TB/s measured float4 copy (SURVEY.md 8d asks for the fraction against both)


def algorithmic_bytes(P, V, F, C):
    """compulsory tensor traffic of one scene, forward + backward."""
    return 20 * P * C + 48 * V + 8 * V * C + 24 * F


def kernel_algorithmic_bytes(P, V, F, C):
    """Share of the 8d figure each kernel is responsible for (DESIGN.md "Kernels"), per scene.
    Set-up reads faces + vertices (counted once per pass that runs it); raster<shade> reads the
    background and vertex colours and writes pixels; raster<visibility> has no algorithmic traffic
    of its own (its output is an intermediate); grad reads pixels, grad_pixels, vertices and writes
    the three gradients."""
    return {
        'setup_kernel': 12 * F + 16 * V,
        'raster_kernel<shade>': 8 * P * C + 4 * V * C,
        'raster_kernel<visibility>': 0,
        'grad_kernel': 12 * P * C + 16 * V + 16 * V + 4 * V * C,
    }


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        return s_.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--scenes-per-gpu', type=int, default=1,
                    help='scenes rendered per rank and step (default 1 at every N: K3 itself per GPU; 8 on 8 GPUs = K4)')
    ap.add_argument('--gather', action='store_true',
                    help='also time gather_batch of every rank\'s pixels to rank 0 (outside the render timing; N > 1)')
    ap.add_argument('--traffic', default='auto', choices=['auto', 'measure', 'file', 'off'],
                    help='roofline.traffic: measured in this run with rocprofv3 --pmc passes (auto: when rocprofv3 is on PATH and N = 1), '
                         'or read from profiles/pmc_traffic.json if its source stamp matches the tree')
    ap.add_argument('--config', default='K3', help='K3 | K3-256 | K3-2048 | K5')
    ap.add_argument('--launch', default='eager', choices=['auto', 'eager', 'graph'],
                    help='how the timed steps are issued: eagerly through the Python wrapper and the C ABI (default: what an autograd user '
                         'pays), as one captured hipGraph replayed per step, or whichever a short calibration finds faster; both are '
                         'reported either way (ms_per_step_eager / ms_per_step_graph)')
    ap.add_argument('--flags', type=lambda x: int(x, 0), default=0, help='extra DIRT_FLAG_* bits for every call (kernel-shape experiments)')
    ap.add_argument('--state-outputs', action='store_true',
                    help='round 4\'s headline: vertex gradients left in the state\'s interleaved accumulators (strided views) instead of the '
                         'op\'s dense contract outputs')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the short K3-256 / K3-2048 legs')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU baseline time budget')
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start N ranks on this node ourselves (one process per GPU,
    # torch.distributed over RCCL); under torch.distributed.run (the driver's way) WORLD_SIZE is already set.
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % args.gpus,
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)')
    # DIRT_BENCH_SHARE_GPU=1: every rank uses GPU 0 and the process group runs over gloo -- ONLY to exercise the N > 1
    # control flow (barriers, max-over-ranks timing, scaling_reference) on a one-GPU box; RCCL refuses two ranks on one
    # device.  The line it prints says so (`config.parallelism`).
    share_gpu = os.environ.get('DIRT_BENCH_SHARE_GPU') == '1'
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    distributed = world > 1
    ranks_seen = 1
    if distributed:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: what RCCL needs on this pool's host driver
        # pre-flight: what this rank sees, before anything that can hang -- and one clear line if the node has fewer GPUs than ranks
        n_dev = torch.cuda.device_count()
        try:
            rccl = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            rccl = 'unavailable (%s)' % type(e).__name__
        if rank == 0 or local_rank >= n_dev:
            print('[bench pre-flight] rank %d/%d local_rank %d: torch.cuda.device_count() = %d, RCCL %s, HSA_ENABLE_IPC_MODE_LEGACY=%s, '
                  'MASTER_ADDR=%s MASTER_PORT=%s, backend %s' % (rank, world, local_rank, n_dev, rccl, os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
                                                                os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'),
                                                                'gloo (DIRT_BENCH_SHARE_GPU)' if share_gpu else 'nccl'), file=sys.stderr, flush=True)
        if not share_gpu and n_dev < int(os.environ.get('LOCAL_WORLD_SIZE', world)):
            raise SystemExit('bench.py --gpus %d: this node exposes %d GPU(s) to rank %d (torch.cuda.device_count()); one rank per GPU is required '
                             '(DIRT_BENCH_SHARE_GPU=1 exercises the control flow over gloo on one GPU)' % (world, n_dev, rank))
        try:
            if share_gpu:
                dist.init_process_group(backend='gloo', timeout=datetime.timedelta(seconds=180))
            else:
                dist.init_process_group(backend='nccl', device_id=dev, timeout=datetime.timedelta(seconds=180))
        except Exception as e:  # noqa: BLE001
            raise SystemExit('bench.py: rank %d could not join the %s process group of %d ranks within 180 s (MASTER_ADDR=%s MASTER_PORT=%s): %s: %s'
                             % (rank, 'gloo' if share_gpu else 'nccl (RCCL)', world, os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'),
                                type(e).__name__, str(e)[:500]))
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)            # proof that RCCL sees every rank; outside the timed region
        ranks_seen = int(ones.item())
        assert ranks_seen == world, 'RCCL all-reduce saw %d of %d ranks' % (ranks_seen, world)

    from dirt_amd import _lib, rasterise_ops as ops
    from tests import scenes
    _lib.load()

    F, H, W, C, seed0, r_lo, r_hi = scenes.CONFIGS[args.config]
    spg = max(1, args.scenes_per_gpu)
    seeds = [seed0 + rank * spg + i for i in range(spg)]
    batch = scenes.batch_scene(F, H, W, C, seeds, r_lo=r_lo, r_hi=r_hi)
    V = batch['vertices'].shape[1]
    P = H * W

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    bg, v, vc, f, g = (t(batch[k]) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels'))

    def step(flags=0):
        flags |= args.flags
        # exactly what torch.autograd does through dirt_amd.rasterise_batch: the forward leaves its
        # set-up records + visibility in a private state buffer, the backward consumes it
        # ... and its launch clears the DENSE grad_vertices [B,V,4] / grad_vertex_colors [B,V,C] that the backward call adds into
        # and returns: the RasteriseGrad op's contract outputs (csrc/rasterise_grad_egl.cpp:381-391), no copy-out launch
        px, state = ops._op_rasterise(bg, v, vc, f, H, W, C, flags=flags, keep_state=True, dense_grads=not args.state_outputs)
        return ops._op_rasterise_grad(v, f, px, g, H, W, C, flags=flags, state=state, state_outputs=True if args.state_outputs else 'dense')

    def barrier():
        if distributed:
            dist.barrier()
        # The host polls an event first: torch.cuda.synchronize() on its own returns ~25 us after the last kernel (the
        # waiting thread has to be woken), which a 20-step timing of a 53 us step would book as 1.2 us per step.  The
        # synchronisation proper follows and finds nothing left to wait for.
        done = torch.cuda.Event()
        done.record()
        while not done.query():
            pass
        torch.cuda.synchronize()

    # Launch mode of the timed loop: the steps issued eagerly through the Python wrapper, or ONE captured hipGraph of the
    # step (set-up + raster + gradient kernels on fixed buffers: the static-shape training-loop deployment) replayed per
    # step -- the same launches either way.  Unless one is asked for, a short calibration inside the warm-up picks the
    # faster (eager while the GPU is the limit, the graph once the host paces the launches).
    def timed(fn, n):
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - c0) / n

    # Graph capture at N > 1 over RCCL is skipped unless asked for (--launch graph | auto): the eager path is the headline
    # either way, the graph figures are reported by the N = 1 run, and capturing while RCCL's watchdog thread polls its
    # events is a known source of "operation not permitted when stream is capturing" -- not a risk to take on a scaling run.
    want_graph = (not distributed) or share_gpu or args.launch != 'eager'
    graph_replay = None
    if want_graph:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, **({'capture_error_mode': 'thread_local'} if distributed else {})):
            graph_out = step()
        graph_replay = graph.replay
    # both ways are timed (a short region each, after a warm-up) and reported; the headline follows --launch
    for _ in range(5):
        step()
        if graph_replay:
            graph_replay()
    n_cal = max(20, min(args.steps, 100))
    calib = {'eager_ms_per_step': timed(step, n_cal) * 1e3,
             'graph_ms_per_step': timed(graph_replay, n_cal) * 1e3 if graph_replay else None, 'steps': n_cal}
    if args.launch == 'auto':
        use_graph = calib['graph_ms_per_step'] < calib['eager_ms_per_step']
        if distributed:  # every rank takes rank 0's choice
            flag = torch.tensor([1 if use_graph else 0], device=dev)
            dist.broadcast(flag, src=0)
            use_graph = bool(flag.item())
    else:
        use_graph = args.launch == 'graph'
    run = graph_replay if use_graph else step

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    ms_per_step = elapsed / args.steps * 1e3
    total_pixels = world * spg * P
    value = total_pixels / (elapsed / args.steps) / 1e6

    # ---- the same K steps between a HIP-event pair ON THE STREAM the kernels are launched on (torch's current
    #      stream: the wrapper passes torch.cuda.current_stream().cuda_stream to the C ABI), no synchronisation between the
    #      warm-up and the timed steps, median of five such regions.  The wall-clock figure above (the contract's: barrier +
    #      synchronize on both sides) also carries the empty-queue ramp of the first steps and the host's wake-up.
    def event_region(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(min(10, n)):
            fn()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n          # ms per step

    regions = sorted(event_region(run, args.steps) for _ in range(5))
    ms_per_step_events = regions[len(regions) // 2]

    # ---- the autograd path: dirt.rasterise_batch(...).backward() with leaf tensors, i.e. what a user of the reference's API
    #      pays -- the same three kernels plus the two copies that hand back DENSE grad_vertices / grad_vertex_colors
    #      (the op returns dense tensors: csrc/rasterise_grad_egl.cpp:381-391) and torch's own bookkeeping ----
    bg_l, v_l, vc_l = (x.clone().requires_grad_(True) for x in (bg, v, vc))

    def autograd_step():
        bg_l.grad = v_l.grad = vc_l.grad = None
        ops.rasterise_batch(bg_l, v_l, vc_l, f, H, W, C).backward(g)

    ms_per_step_autograd = sorted(event_region(autograd_step, max(20, min(args.steps, 200))) for _ in range(3))[1]
    # ... and with torch's backward engine kept on the calling thread (torch.autograd.set_multithreading_enabled(False): a
    # user-side switch): the engine otherwise hands every CUDA node to a per-device worker thread and waits for it -- two
    # thread wake-ups per backward(), which is most of what the autograd path costs over the raw ops at this step length
    with torch.autograd.set_multithreading_enabled(False):
        ms_per_step_autograd_1t = sorted(event_region(autograd_step, max(20, min(args.steps, 200))) for _ in range(3))[1]
    # ... and the remedy that needs no switch: the same forward + backward captured once as a HIP graph
    # (dirt_amd.GraphedStep: rasterise_batch -> backward(grad_pixels) recorded on fixed buffers, one hipGraphLaunch per step,
    # dense gradients) -- what a static-shape training loop should call
    ms_per_step_autograd_graphed = None
    try:
        if not want_graph:
            raise RuntimeError('skipped at N > 1 (reported by the N = 1 run)')
        from dirt_amd import GraphedStep
        gstep = GraphedStep(bg_l.detach(), v_l.detach(), vc_l.detach(), f, grad_pixels=g)
        ms_per_step_autograd_graphed = sorted(event_region(gstep, max(20, min(args.steps, 200))) for _ in range(3))[1]
        del gstep
    except Exception as e:   # reported, never fatal to the bench line
        ms_per_step_autograd_graphed = 'failed: %s' % (str(e)[:100],)
    del bg_l, v_l, vc_l

    # ---- the other frame sizes BASELINE.json's north_star asks for (256^2 and 2048^2, same mesh): short legs, one scene ----
    other_configs = {}
    if args.config == 'K3' and not args.no_other_configs:
        # (at N > 1 every rank renders its own scene of the leg at the same time, as in the headline: the figure is the
        # max over ranks, the value the pixels of all ranks over it)
        for name in ('K3-256', 'K3-2048'):
            F2, H2, W2, C2, seed2, lo2, hi2 = scenes.CONFIGS[name]
            b2 = scenes.batch_scene(F2, H2, W2, C2, [seed2 + rank], r_lo=lo2, r_hi=hi2)
            bg2, v2, vc2, f2, g2 = (t(b2[k]) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels'))

            def step2():
                px2, st2 = ops._op_rasterise(bg2, v2, vc2, f2, H2, W2, C2, flags=args.flags, keep_state=True, dense_grads=True)
                return ops._op_rasterise_grad(v2, f2, px2, g2, H2, W2, C2, flags=args.flags, state=st2, state_outputs='dense')

            if distributed:
                dist.barrier()
            ms2 = sorted(event_region(step2, 100) for _ in range(3))[1]
            if distributed:
                tm = torch.tensor([ms2], dtype=torch.float64, device=dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                ms2 = float(tm.item())
            by2 = algorithmic_bytes(H2 * W2, b2['vertices'].shape[1], F2, C2)
            other_configs[name] = {'ms_per_step': ms2, 'value': world * H2 * W2 / ms2 / 1e3, 'unit': 'Mpixels/s', 'steps': 100, 'n_gpus': world,
                                   'roofline_step_frac': by2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                   'workload': 'rand_mesh F=%d at %dx%dx%d, 1 scene per GPU, forward+backward, dense outputs' % (F2, H2, W2, C2)}
            del bg2, v2, vc2, f2, g2

    # ---- N > 1: the same per-GPU workload on rank 0 ALONE (every other rank idles at the barrier), so that the line
    #      carries its own one-GPU reference for exactly this scenes-per-GPU count ----
    scaling_reference = None
    if distributed:
        dist.barrier()
        if rank == 0:
            for _ in range(min(args.warmup, 20)):
                run()
            n_ref = max(20, min(args.steps, 200))
            t_ref = timed(run, n_ref)
            scaling_reference = {'n_gpus': 1, 'scenes_per_gpu': spg, 'steps': n_ref, 'ms_per_step': t_ref * 1e3,
                                 'value': spg * P / t_ref / 1e6, 'unit': 'Mpixels/s',
                                 'note': 'rank 0 alone, the other ranks idle; value / (n_gpus * this) is the weak-scaling efficiency'}
        dist.barrier()

    # ---- optional: collecting every rank's pixels on rank 0 (grouped send / recv over RCCL, dirt_amd/sharding.py), timed
    #      on its own -- it is per-link bound over xGMI and never part of the render timing ----
    gather = None
    if args.gather and distributed:
        from dirt_amd import sharding
        px = ops._op_rasterise(bg, v, vc, f, H, W, C)
        for _ in range(2):
            sharding.gather_batch(px, world * spg, 0)
        barrier()
        n_g = 10
        c0 = time.perf_counter()
        for _ in range(n_g):
            full = sharding.gather_batch(px, world * spg, 0)
        barrier()
        t_g = (time.perf_counter() - c0) / n_g
        tg = torch.tensor([t_g], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        nbytes = (world - 1) * spg * P * C * 4
        gather = {'what': 'pixels of all %d scenes to rank 0' % (world * spg), 'ms': float(tg.item()) * 1e3, 'bytes_received': nbytes,
                  'GB_per_s': nbytes / float(tg.item()) / 1e9,
                  'render_plus_gather_value': total_pixels / (elapsed / args.steps + float(tg.item())) / 1e6,
                  'rank0_has_all': bool(rank != 0 or (full is not None and full.shape[0] == world * spg))}

    # ---- per-kernel HIP-event timing over a second, identical timed region (rank 0 only) ----
    roofline = None
    kernels = {}
    if rank == 0:
        _lib.profile_reset()
        torch.cuda.synchronize()
        for _ in range(args.steps):
            step(_lib.FLAG_PROFILE)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        kbytes = kernel_algorithmic_bytes(P, V, F, C)
        # avg_us is the RAW reading of the event pair around a launch.  An event pair reads ~2 us longer than the kernel runs
        # (the closing event's packet is processed after the kernel has drained), so the raw readings of a step sum to more
        # than the step takes; `avg_us_corrected` takes that excess, shared equally among the step's launches, off -- an
        # estimate (it agrees with rocprofv3's kernel trace within ~3 %, profiles/README.md), never used for the roofline.
        raw_us = {name: ms / n * 1e3 for name, (ms, n) in prof.items() if n}
        launches = sum(n for _, n in prof.values()) / args.steps
        eager_step_us = (ms_per_step if not use_graph else calib['eager_ms_per_step']) * 1e3
        sum_raw = sum(ms for ms, _ in prof.values()) / args.steps * 1e3
        pair_us = max(0.0, (sum_raw - eager_step_us) / launches) if launches else 0.0
        for name, (ms, n) in prof.items():
            if n:
                raw = raw_us[name]
                kernels[name] = {'launches_per_step': n / args.steps, 'avg_us': raw, 'avg_us_corrected': max(raw - pair_us, 0.0),
                                 'us_per_step': raw * n / args.steps}
        dom = max(kernels, key=lambda k: kernels[k]['us_per_step'])
        avg_s = kernels[dom]['avg_us'] * 1e-6
        per_launch = kbytes[dom] * spg
        achieved = per_launch / avg_s / 1e9
        traffic, traffic_source, traffic_all = None, None, None
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import measure_traffic
        want_measure = args.traffic == 'measure' or (args.traffic == 'auto' and world == 1 and spg == 1)
        if want_measure:
            try:
                traffic_all = measure_traffic.measure(args.config)
                traffic = traffic_all.get(dom)
                traffic_source = ('measured in this run: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes) over 3 steps of this '
                                  'workload, (2 * FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 correction of MI355X_MICROARCH.md)')
            except Exception as e:  # no rocprofv3, no permission, time-out: fall back to the stamped file
                traffic_source = 'in-run measurement failed (%s)' % (str(e)[:80],)
        if traffic is None and args.traffic != 'off':
            pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')  # written by tools/measure_traffic.py
            try:
                pj = json.load(open(pmc)).get(args.config, {})
                if pj.get('_sources_sha256') == measure_traffic.source_stamp():
                    traffic = pj.get(dom)
                    if traffic is not None:
                        traffic *= spg   # counted on one scene per launch; a launch of this run renders `spg` independent scenes
                        traffic_source = ((traffic_source + '; ' if traffic_source else '') +
                                          'profiles/pmc_traffic.json, collected on these very kernel sources (sha256 stamp matches)'
                                          + (' x %d scenes per launch' % spg if spg > 1 else ''))
                else:
                    traffic_source = (traffic_source + '; ' if traffic_source else '') + 'profiles/pmc_traffic.json is stale (source stamp differs): not used'
            except Exception:
                pass
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                    'frac': achieved / HBM_PEAK_GBPS, 'frac_of_measured_copy_peak': achieved / HBM_COPY_GBPS,
                    'traffic': traffic, 'traffic_source': traffic_source, 'traffic_all_kernels': traffic_all,
                    'algorithmic_bytes_per_launch': per_launch, 'avg_launch_us': kernels[dom]['avg_us'],
                    'avg_launch_us_corrected': kernels[dom]['avg_us_corrected'], 'event_pair_us': pair_us,
                    'frac_corrected': per_launch / (kernels[dom]['avg_us_corrected'] * 1e-6) / 1e9 / HBM_PEAK_GBPS if kernels[dom]['avg_us_corrected'] else None,
                    'timing': 'achieved / frac: the RAW HIP-event reading around each launch on its stream (DIRT_FLAG_PROFILE, eager steps), '
                              'which includes ~2 us of event-pair overhead -- conservative; *_corrected takes (sum of the raw readings of a '
                              'step - the eager step time) / launches per step off, which agrees with the rocprofv3 kernel trace of the same '
                              'command within ~3 % (profiles/)'}

    # ---- CPU baseline: the oracle on this host's cores, bounded sample (rank 0, N == 1) ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        oracle.build()
        one = {k: batch[k][:1] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        # choose the thread count that is fastest on this host (all cores is not: the gradient
        # scatter contends); one untimed forward+backward per candidate
        best, cores = None, 1
        for n in sorted({avail, max(1, avail // 2), 64, 32, 16, 8}):
            if n > avail:
                continue
            oracle.set_num_threads(n)
            c0 = time.perf_counter()
            px = oracle.forward(one['background'], one['vertices'], one['vertex_colors'], one['faces'])
            oracle.backward(one['vertices'], one['faces'], px, one['grad_pixels'])
            dt = time.perf_counter() - c0
            if best is None or dt < best:
                best, cores = dt, n
        oracle.set_num_threads(cores)
        n_it, t_cpu = 0, 0.0
        while t_cpu < args.cpu_seconds and n_it < 400:
            c0 = time.perf_counter()
            px = oracle.forward(one['background'], one['vertices'], one['vertex_colors'], one['faces'])
            oracle.backward(one['vertices'], one['faces'], px, one['grad_pixels'])
            t_cpu += time.perf_counter() - c0
            n_it += 1
        cpu_baseline = {'value': n_it * P / t_cpu / 1e6, 'unit': 'Mpixels/s', 'cores': oracle.num_threads(),
                        'kind': 'port',
                        'sample': '%d x forward+backward of one %s scene (%dx%dx%d, %d triangles), oracle/dirt_oracle.c with OpenMP'
                                  % (n_it, args.config, H, W, C, F)}
        # "single-thread and all-core": the same oracle on ONE host thread, two passes
        oracle.set_num_threads(1)
        n_1, t_1 = 0, 0.0
        while t_1 < 3.0 and n_1 < 3:
            c0 = time.perf_counter()
            px = oracle.forward(one['background'], one['vertices'], one['vertex_colors'], one['faces'])
            oracle.backward(one['vertices'], one['faces'], px, one['grad_pixels'])
            t_1 += time.perf_counter() - c0
            n_1 += 1
        oracle.set_num_threads(cores)
        cpu_baseline['single_thread'] = {'value': n_1 * P / t_1 / 1e6, 'unit': 'Mpixels/s', 'cores': 1, 'kind': 'port',
                                         'sample': '%d x forward+backward of the same scene on one thread' % n_1}
        cpu_baseline['host_threads_available'] = avail
        # The one part of the reference that compiles for a CPU -- its gradient kernel assemble_grads + launch_grad_assembly,
        # csrc/rasterise_grad_egl.cu, behind oracle/ref_shim (oracle/_ref, prebuilt: it travels with the snapshot) -- timed
        # beside it: BACKWARD only (the channel groups of dirt/rasterise_ops.py:145-165, surfaces from the oracle's
        # visibility, which is not timed), one thread by construction (the shim runs the CUDA grid as one thread).
        try:
            from oracle import ref as _ref
            if _ref.available():
                surf = _ref.surfaces(one['vertices'], one['faces'], H, W)
                px1 = oracle.forward(one['background'], one['vertices'], one['vertex_colors'], one['faces'])

                def ref_backward():
                    begin = 0
                    while begin < C:
                        end = begin + 3 if begin + 3 <= C else begin + 1
                        _ref.rasterise_grad_op(one['vertices'], one['faces'], px1[..., begin:end], one['grad_pixels'][..., begin:end], surf)
                        begin = end
                ref_backward()
                n_r, t_r = 0, 0.0
                while t_r < 3.0 and n_r < 50:
                    c0 = time.perf_counter()
                    ref_backward()
                    t_r += time.perf_counter() - c0
                    n_r += 1
                cpu_baseline['reference_kernel_backward'] = {
                    'value': n_r * P / t_r / 1e6, 'unit': 'Mpixels/s (backward only)', 'cores': 1, 'kind': 'reference',
                    'sample': '%d x the reference\'s own assemble_grads (csrc/rasterise_grad_egl.cu compiled for the host, oracle/_ref) over '
                              'the %d channel groups of one %s scene' % (n_r, (C // 3) + (C % 3), args.config)}
        except Exception as e:  # the checker is optional here: never let it take the bench line down
            cpu_baseline['reference_kernel_backward'] = {'error': str(e)[:120]}

    if rank == 0:
        step_bytes = algorithmic_bytes(P, V, F, C) * spg
        out = {
            'metric': 'forward+backward Mpixels/s at 1024x1024x4ch, 10k-tri mesh' if args.config == 'K3'
                      else 'forward+backward Mpixels/s (%s)' % args.config,
            'value': value, 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (f64 edge functions)', 'data': 'synthetic',
            'config': {'workload': '%s: rand_mesh F=%d at %dx%dx%d, %d scene(s) per GPU, forward+backward'
                                   % (args.config, F, H, W, C, spg),
                       'outputs': 'strided views of the state\'s interleaved accumulators (--state-outputs)' if args.state_outputs else
                                  'dense grad_background [B,H,W,C], grad_vertices [B,V,4], grad_vertex_colors [B,V,C] (the op\'s contract outputs; '
                                  'cleared inside the forward launch, no copy-out launch)',
                       'scenes_per_gpu': spg, 'parallelism': 'batch-sharded x%d, no collective' % world + (' (DIRT_BENCH_SHARE_GPU: all ranks on ONE GPU over gloo, control-flow test only)' if share_gpu else ''),
                       'launch': 'one captured hipGraph replayed per step' if use_graph else 'eager (Python wrapper + C ABI per step)'},
            'ranks_seen': ranks_seen,
            'ms_per_step_events_median': ms_per_step_events,
            'value_events_median': total_pixels / ms_per_step_events / 1e3,
            'timing': 'ms_per_step / value: wall clock over the K steps between barrier + synchronize (the contract); '
                      'ms_per_step_events_median: the same K steps between a HIP-event pair on the launch stream, no synchronisation '
                      'after the warm-up, median of 5 regions (SURVEY.md 8d); ms_per_step_autograd: dirt.rasterise_batch(...).backward() '
                      'with leaf tensors and dense gradients, event-timed; ..._engine_on_calling_thread: the same under '
                      'torch.autograd.set_multithreading_enabled(False) (no hand-off to the per-device backward thread); ..._graphed: the same '
                      'rasterise_batch -> backward captured once as a HIP graph (dirt_amd.GraphedStep), one graph launch per step',
            'ms_per_step_autograd': ms_per_step_autograd,
            'ms_per_step_autograd_engine_on_calling_thread': ms_per_step_autograd_1t,
            'ms_per_step_autograd_graphed': ms_per_step_autograd_graphed,
            'other_configs': other_configs,
            'ms_per_step_eager': calib['eager_ms_per_step'], 'ms_per_step_graph': calib['graph_ms_per_step'],
            'launch_calibration': calib,
            'scaling_reference': scaling_reference,
            'gather': gather,
            'roofline': roofline,
            'roofline_step': {'algorithmic_bytes': step_bytes, 'achieved': step_bytes / (ms_per_step * 1e-3) / 1e9,
                              'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                              'frac': step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              'frac_of_measured_copy_peak': step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_COPY_GBPS},
            'kernels': kernels,
            'cpu_baseline': cpu_baseline,
        }
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
