"""bench.py -- clips/s of the SlowFastLayers hot path (fwd + bwd + SGD) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
torch.distributed.run with one rank per GPU (RCCL).  Started as a plain `python bench.py --gpus N` (no WORLD_SIZE in
the environment) it starts the N ranks itself -- as a CHILD `python -m torch.distributed.run ...` process, before
this process has made any GPU call -- and exits with the child's status.  Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[1] restated in the reference's own parametrisation:
(slow, fast) = (4, 32) frames, one clip = one call of temporally_enhance_features over the five FPN levels of a
480x854 DAVIS frame (P = 85 932 positions), bf16 activations / f32 accumulate, the clip resident in HBM as a
channel-group-major bf16 PackedClip ([C/32][positions][32]: SURVEY.md 8f.3 hand-over), synthetic N(0,1) features,
random-init weights.  A step = forward + loss + backward of one clip; gradients are accumulated over 2 clips and
then the optimiser steps (reference model.py:369-374), with the data-parallel all-reduce in front of it.
The loss is the stand-in of SURVEY.md 8d (sum_l mean((out_l - target_l)^2), sfvos_amd.MSEProxyLoss: two libsfvos
launches); nothing under oracle/ is imported outside cpu_baseline().
Streams: the timed region uses the module's default of two HIP streams per clip (slow pathway + laterals on a side
stream, which starts BEHIND fast_conv1's forward launch: the dominant kernel runs alone, the pathways then fill each
other's launch tails; `--streams 1` serialises everything: 4 % slower).
HIP events: inside the timed region only the dominant kernel (fast_conv1 forward) is bracketed -> `roofline`; the
per-layer tables (`mfma_layers`, `hbm_layers`, `hbm_bound_passes`, `kernels_ms`) come from a second, untimed pass of
the same step on ONE stream with every launch bracketed (`--kernel-events all` puts them back into the timed region).
`dropin_api_ms_per_step` times the same step through the reference's calling convention
(temporally_enhance_features on lists of fp32 NCHW frame tensors, model.py:157-158,340), i.e. including the
fp32-NCHW -> bf16 layout pass that a train.py caller pays on every call.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--sp', type=int, default=4)
    ap.add_argument('--fp', type=int, default=32)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32', 'fp8'],
                    help="fp8 = BASELINE config 5: eval-mode FORWARD only (the e4m3 path has no backward), its own metric "
                         "name so that the line cannot be mistaken for the headline")
    ap.add_argument('--kernel-events', choices=['all', 'dominant', 'none'], default='dominant',
                    help='HIP events in the timed region: around every launch / only the dominant kernel / none')
    ap.add_argument('--no-layer-table', action='store_true',
                    help='skip the per-layer pass (every launch bracketed by events) behind the timed region')
    ap.add_argument('--overlap-f1', action='store_true',
                    help='with --streams 2: let the side stream start alongside fast_conv1 forward (A/B)')
    ap.add_argument('--streams', type=int, default=2,
                    help='HIP streams per clip in the timed region: 2 (the module\'s default) = slow pathway + laterals '
                         'on a side stream that starts BEHIND fast_conv1\'s forward launch, so the dominant kernel still '
                         'runs alone and its HIP-event duration is clean; 1 = every launch on one stream (the '
                         'per-layer pass always uses 1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dropin', action='store_true', help='skip the drop-in API (fp32 NCHW frame lists) timing')
    ap.add_argument('--ndhwc-input', action='store_true', help='hand the clip over as plain pyramid NDHWC (A/B of the grouped layout)')
    ap.add_argument('--no-grad-sink', action='store_true',
                    help='let autograd accumulate parameter gradients from temporaries (A/B of FusedSGD.attach)')
    ap.add_argument('--per-layer-allreduce', action='store_true',
                    help='one all-reduce per layer instead of the 4 coalesced buckets (A/B)')
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--cpu-clips', type=int, default=5, help='timed clips of the CPU baseline (after 1 warm-up)')
    ap.add_argument('--graph', action='store_true',
                    help='replay the accumulate-2 cycle from hipGraphs (sfvos_amd.GraphedStep) instead of launching it '
                         'from Python: same kernels, same order; pays on launch-bound configurations such as the '
                         "reference's default --sp 1 --fp 1 (one rank only; the dominant kernel's duration then comes "
                         'from the per-layer pass, events cannot bracket a launch inside a graph)')
    ap.add_argument('--force-dist', action='store_true',
                    help='create the RCCL (nccl) process group even with ONE rank and send the gradient buckets through it '
                         '(an all-reduce over one rank is the identity): the RCCL path of the exchange on a single GPU')
    ap.add_argument('--master-port', type=int, default=29541)
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as a child process.
    Nothing in this process has touched the GPU (torch is not even imported yet)."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(sp, fp, threads, clips):
    """The oracle (torch-CPU restatement of the reference path, parity-locked to the reference's class by
    tests/golden) timed on this box's host cores: whole clips (all five FPN levels through
    temporally_enhance_features, B = 1), fwd + loss + bwd, SGD(lr 1e-3, momentum 0.9, wd 1e-4) every 2nd clip as
    model.py:372-374 steps it; 1 warm-up clip, then the median of `clips` timed clips."""
    import torch
    from collections import OrderedDict
    from oracle.slowfast_ref import OracleSlowFastLayers, proxy_loss
    from sfvos_amd import davis_pyramid
    if threads > 0:
        torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    pyr = davis_pyramid()
    torch.manual_seed(63)
    m = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    fast = OrderedDict((k, torch.randn(fp, 256, h, w)) for k, (h, w) in pyr)
    idx = fp // 2
    slow = OrderedDict((k, v[idx - sp // 2: idx + (sp + 1) // 2]) for k, v in fast.items())
    times = []
    for i in range(1 + clips):
        t0 = time.perf_counter()
        loss = proxy_loss(m.temporally_enhance_features([slow], [fast]))
        loss.backward()
        if i % 2 == 1:
            opt.step()
            opt.zero_grad()
        times.append(time.perf_counter() - t0)
    med = statistics.median(times[1:])
    return {'value': 1.0 / med, 'unit': 'clips/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(),
            's_per_clip': round(med, 3),
            'sample': 'oracle (torch fp32 CPU) fwd+loss+bwd of whole (sp=%d, fp=%d) clips over the 5-level DAVIS '
                      'pyramid, SGD every 2nd clip; 1 warm-up (%.1f s) + median of %d timed clips (%s s), %d threads'
                      % (sp, fp, times[0], clips, ', '.join('%.1f' % t for t in times[1:]), cores)}


def pmc_traffic(path, kernel):
    """HBM bytes of the dominant kernel per launch, from the committed rocprofv3 PMC passes of this same
    command (FETCH_SIZE and WRITE_SIZE in separate passes; KB units; gfx950 FETCH_SIZE counts half of a wide
    streaming read, so it is doubled -- MI355X_MICROARCH.md, HBM).  None when the summary is absent."""
    try:
        with open(path) as f:
            s = json.load(f)
        return 2.0 * s[kernel]['FETCH_SIZE']['mean'] * 1024 + s[kernel]['WRITE_SIZE']['mean'] * 1024
    except Exception:
        return None


PMC_SUMMARY = 'r03_pmc_summary.json'   # the committed counter summary `roofline.traffic` is read from


def make_step(model, opt, bucket, loss_fn, forward):
    """One bench step (also driven, at world size 2 over gloo, by tests/test_gpu_dp.py): forward + loss + backward of
    one clip; every 2nd clip completes the gradients: all-reduce (overlapped with that backward), optimiser step."""
    def step(i):
        loss = loss_fn(forward())
        if i % 2 == 1:  # model.py:372-374: optimiser every 2nd clip: this backward completes the gradients
            bucket.arm()
        loss.backward()
        if i % 2 == 1:
            bucket.finish()
            opt.step()
            opt.zero_grad()
        return loss
    return step


def main_fp8(args):
    """BASELINE config 5: the e4m3 conv path, eval-mode FORWARD of one (sp, fp) clip over the DAVIS
    pyramid -- fast_conv1 and slow_conv1-3 (99 % of the forward FLOPs) on v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3
    operands, f32 accumulate, bf16 results; the 32-channel layers stay bf16.  The clip is resident in HBM as e4m3
    64-channel groups (SlowFastLayers.pack_fp8).  Not the headline metric: no backward exists in e4m3, so the line
    carries its own metric name.  The bf16 eval forward of the same clip is timed beside it."""
    import torch
    from sfvos_amd import PackedClip, SlowFastLayers, davis_pyramid
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(63)
    pyr = davis_pyramid()
    P = sum(h * w for _, (h, w) in pyr)
    gen = torch.Generator(device=dev).manual_seed(63)
    model = SlowFastLayers(256, dev, args.sp, args.fp, precision='fp8').to(dev)
    model.eval()
    model.n_streams = args.streams
    from collections import OrderedDict
    fast = OrderedDict((k, torch.randn((args.fp, 256, h, w), generator=gen, device=dev)) for k, (h, w) in pyr)
    model.calibrate_fp8_scale([fast])
    clip8 = model.pack_fp8([fast])
    levels = [v.permute(0, 2, 3, 1).unsqueeze(0).to(torch.bfloat16) for v in fast.values()]
    clip16 = PackedClip.from_levels(levels, keys=[k for k, _ in pyr], layout='grouped')
    del levels, fast
    sat = model.fp8_saturated()

    def run(precision, clip, steps, warmup, only):
        model.precision = precision
        timer = model.enable_kernel_timer(only)
        with torch.no_grad():
            for _ in range(warmup):
                model.enhance_packed(clip)
            timer.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                model.enhance_packed(clip)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        k = timer.summary()
        model._timer = None
        return dt, k

    dt, kdom = run('fp8', clip8, args.steps, args.warmup, ['conv_fwd/f1'])
    model.n_streams = 1
    _, k8 = run('fp8', clip8, 6, 2, None)
    _, k16 = run('bf16', clip16, 6, 2, None)
    model.n_streams = args.streams
    dt16, _ = run('bf16', clip16, args.steps, args.warmup, [])
    plan = model.plan
    l = plan.layer('f1')
    fl = plan.layer_flops(P)
    dom = kdom['conv_fwd/f1']
    ach = fl['f1'] / (dom[1] * 1e-3) / 1e12
    layers = {}
    for name in ('s1', 'f1', 's2', 's3'):
        a, b = k8.get('conv_fwd/' + name), k16.get('conv_fwd/' + name)
        if a and b:
            layers[name] = {'fp8_ms': round(a[1], 4), 'bf16_ms': round(b[1], 4),
                            'fp8_TFLOPs': round(fl[name] / a[1] / 1e9, 1), 'frac_of_5PF': round(fl[name] / a[1] / 1e9 / 5000.0, 3)}
    line = {
        'metric': 'clips/sec (T=%d, 480x854) eval FORWARD only, e4m3 conv operands (BASELINE config 5; not the headline fwd+bwd metric)' % args.fp,
        'value': round(args.steps / dt, 3), 'unit': 'clips/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'fp8 (e4m3 operands of the four Cin=256 convs, f32 accumulate, bf16 results; 32-channel layers bf16)',
        'data': 'synthetic',
        'config': {'workload': 'SlowFastLayers (sp=%d, fp=%d) eval forward, 1 clip/step, 5 FPN levels of a 480x854 frame '
                               '(P=%d); input = e4m3 clip resident in HBM ([C/64][pos][64] groups, per-tensor scale %.3g, '
                               '%d of %d elements saturated)' % (args.sp, args.fp, P, model.fp8_input_scale, sat,
                                                                 args.fp * P * 256),
                   'parallelism': 'dp1', 'hip_streams_per_clip': args.streams},
        'bf16_forward_ms_per_step': round(1e3 * dt16 / args.steps, 3),
        'fp8_over_bf16_forward': round(dt16 / dt, 3),
        'tflops_per_clip_forward': round(plan.forward_flops(P) / 1e12, 3),
        'roofline': {'bound': 'mfma',
                     'kernel': 'sfvos::conv3d_fs_kernel<2,256> (fast_conv1 forward on e4m3 operands, 256->32 ch, %dx3x3, '
                               '%d->%d frames, one launch)' % (l.kt, l.t_in, l.t_out),
                     'achieved': round(ach, 2), 'peak': 5000.0, 'unit': 'TFLOP/s', 'frac': round(ach / 5000.0, 4),
                     'launch_ms': round(dom[1], 4), 'flops_per_launch': fl['f1'], 'traffic': None},
        'fp8_layers': layers,
        'kernels_ms': {k: [v[0], round(v[1], 4)] for k, v in sorted(k8.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:20]},
        'cpu_baseline': None,
        'note': 'tolerance of this path (tests/test_gpu_parity.py::test_fp8_inference_path_error_is_measured_and_bounded): '
                'fused maps 5 % rel-L2 against the fp32 oracle, argmax agreement 92-98 %',
    }
    print(json.dumps(line))
    sys.stdout.flush()


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist
    from sfvos_amd import (FusedSGD, GradBucket, MSEProxyLoss, PackedClip, SlowFastLayers, davis_pyramid,
                           init_distributed)

    if args.precision == 'fp8':
        return main_fp8(args)
    rank, world, local = init_distributed(force=args.force_dist)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        # RCCL prints a version banner on stdout when its communicator is created (lazily, at the first collective):
        # create it now with stdout pointed at stderr, so that stdout carries nothing but rank 0's ONE JSON line
        torch.cuda.set_device((local % torch.cuda.device_count()) if world > 1 else 0)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    if world != args.gpus and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)
    dev = torch.device('cuda', (local % torch.cuda.device_count()) if world > 1 else 0)
    torch.cuda.set_device(dev)

    torch.manual_seed(63)
    model = SlowFastLayers(256, dev, args.sp, args.fp, precision=args.precision).to(dev)
    model.train()
    model.n_streams = args.streams
    model.f1_alone = not args.overlap_f1
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    bucket = GradBucket(opt.flat_grad, coalesce=not args.per_layer_allreduce, force=args.force_dist)
    if not args.no_grad_sink:
        # kernels write parameter gradients straight into the flat gradient buffer; with world > 1 the finished
        # layers' slices are all-reduced (RCCL, side stream) while backward continues
        opt.attach(model, bucket)
    tdt = torch.bfloat16 if args.precision == 'bf16' else torch.float32
    pyr = davis_pyramid()
    P = sum(h * w for _, (h, w) in pyr)
    gen = torch.Generator(device=dev).manual_seed(63 + rank)
    # one clip = the fast window of every FPN level, channels-last
    levels = [torch.randn((1, args.fp, h, w, 256), generator=gen, device=dev, dtype=torch.float32).to(tdt)
              for _, (h, w) in pyr]
    grouped = args.precision == 'bf16' and not args.ndhwc_input
    clip = PackedClip.from_levels(levels, keys=[k for k, _ in pyr], layout='grouped' if grouped else 'ndhwc')
    del levels
    loss_fn = MSEProxyLoss({k: torch.randn((1, 256, h, w), generator=gen, device=dev) for k, (h, w) in pyr})

    # HIP events in the timed region: around the dominant kernel only (roofline.launch_ms).  An event record between
    # two kernels costs about a microsecond of GPU time and a step has ~100 of them: with every launch bracketed the
    # step is 3 % slower (A/B on one box: 9.70 vs 9.40 ms), so the per-layer table comes from a separate pass below.
    if args.graph and (dist_on or args.kernel_events == 'all'):
        sys.exit('--graph: one rank without --force-dist, and not with --kernel-events all')
    timer = model.enable_kernel_timer(None if args.kernel_events == 'all' else
                                      (['conv_fwd/f1'] if args.kernel_events == 'dominant' and not args.graph else []))
    step = make_step(model, opt, bucket, loss_fn, lambda: model.enhance_packed(clip))
    run = step
    if args.graph:
        from sfvos_amd import GraphedStep
        gstep = GraphedStep(model, opt, loss_fn, clip, accumulate=2, bucket=bucket)
        run = lambda i: gstep()   # noqa: E731  (steps and warm-up are even: whole cycles)
        if args.warmup % 2 or args.steps % 2:
            sys.exit('--graph: --steps and --warmup must be even (whole accumulate-2 cycles)')

    for i in range(args.warmup):
        run(i)
    timer.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_dom = timer.summary()  # name -> (calls, mean ms): the timed region's own events
    kern, table_steps = kern_dom, args.steps
    if args.kernel_events != 'all' and not args.no_layer_table:
        # per-layer table: the same step with every launch bracketed, outside the timed region
        table_steps = max(2, min(args.steps, 10)) // 2 * 2
        model.n_streams = 1   # one stream: every kernel runs alone, its event duration is its own
        timer = model.enable_kernel_timer()
        for i in range(table_steps):
            step(i)
        kern = timer.summary()
        model.n_streams = args.streams
    model._timer = None

    # ---- the same step through the reference's calling convention: lists of fp32 NCHW frames -> layout pass inside
    dropin_ms = None
    if not args.no_dropin and world == 1:
        from collections import OrderedDict
        fast = OrderedDict((k, torch.randn((args.fp, 256, h, w), generator=gen, device=dev)) for k, (h, w) in pyr)
        idx = args.fp // 2
        slow = OrderedDict((k, v[idx - args.sp // 2: idx + (args.sp + 1) // 2]) for k, v in fast.items())
        dstep = make_step(model, opt, bucket, loss_fn, lambda: model.temporally_enhance_features([slow], [fast]))
        n = max(2, min(args.steps, 10))
        for i in range(2):
            dstep(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n):
            dstep(i)
        torch.cuda.synchronize()
        dropin_ms = 1e3 * (time.perf_counter() - t1) / n
        del fast, slow

    if rank == 0:
        plan = model.plan
        # dominant kernel: fast_conv1 forward over the whole 5-level pyramid (72 % of the forward
        # FLOPs): ONE launch of conv3d_fs_kernel<bf16, 256> holding the frame blocks of 4, 2 and 1 output frames;
        # the HIP events (on the stream it is launched on) bracket exactly that launch.
        l = plan.layer('f1')
        dom_flops = 2.0 * l.c_in * l.c_out * l.kt * l.taps * l.t_out * P
        dom = kern_dom.get('conv_fwd/f1') or kern.get('conv_fwd/f1')
        peak = 2500.0 if args.precision == 'bf16' else 157.3
        roofline = None
        if dom:
            ach = dom_flops / (dom[1] * 1e-3) / 1e12
            headline = (args.sp, args.fp, args.precision) == (4, 32, 'bf16')
            roofline = {'bound': 'mfma',
                        'kernel': 'sfvos::conv3d_fs_kernel<%s,256> (fast_conv1 forward, 256->32 ch, %dx3x3, '
                                  '%d->%d frames, 5-level pyramid, one launch)'
                                  % ('1' if args.precision == 'bf16' else '0', l.kt, l.t_in, l.t_out),
                        'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                        'launch_ms': round(dom[1], 4), 'flops_per_launch': dom_flops,
                        'traffic': pmc_traffic(os.path.join(ROOT, 'profiles', PMC_SUMMARY),
                                               'conv3d_fs_kernel<1, 256>') if headline else None,
                        'traffic_source': ('profiles/%s (static: rocprofv3 --pmc passes of this command, committed; '
                                           'not re-measured in this run)' % PMC_SUMMARY) if headline else None}
        # HBM-bound passes: algorithmic bytes (each tensor touched once per pass) / HIP-event time
        es = 2 if args.precision == 'bf16' else 4
        hbm = {}
        for kind, passes in (('bn_apply', 2), ('bn_bwd', 5)):   # apply: read x, write y; bwd: 2x(dy, x) + write dx
            nbytes = sum(l.t_out * P * l.c_out * es * passes for l in plan.layers)
            ms = sum(kern[k][1] for k in kern if k.startswith(kind + '/'))
            if ms > 0:
                hbm[kind] = {'gbytes': round(nbytes / 1e9, 3), 'ms': round(ms, 4),
                             'achieved_GBps': round(nbytes / ms / 1e6, 1), 'peak_GBps': 8000.0,
                             'frac': round(nbytes / ms / 1e6 / 8000.0, 4)}
        mfma = {}
        fl = plan.layer_flops(P)
        for kind in ('conv_fwd', 'wgrad', 'conv_dgrad'):
            for l in plan.layers:
                k = '%s/%s' % (kind, l.name)
                if k in kern:
                    tf = fl[l.name] / kern[k][1] / 1e9
                    mfma[k] = {'ms': round(kern[k][1], 4), 'TFLOPs': round(tf, 1), 'frac': round(tf / peak, 3)}
        # HBM-bound conv layers (laterals AI 88-137 F/B, fast_conv3 265): algorithmic bytes / event time.
        # forward: read x, write y; weight gradient: read x and dy; data gradient: read dy, read + write dx (it
        # accumulates onto the other consumer's gradient)
        hbm_layers = {}
        for l in plan.layers:
            if l.name not in ('l1', 'l2', 'f3'):
                continue
            xin, yout = l.t_in * P * l.c_in * es, l.t_out * P * l.c_out * es
            for kind, nbytes in (('conv_fwd', xin + yout), ('wgrad', xin + yout), ('conv_dgrad', yout + 2 * xin)):
                k = '%s/%s' % (kind, l.name)
                if k in kern:
                    bw = nbytes / kern[k][1] / 1e6
                    hbm_layers[k] = {'ms': round(kern[k][1], 4), 'gbytes': round(nbytes / 1e9, 4),
                                     'achieved_GBps': round(bw, 1), 'frac': round(bw / 8000.0, 3)}
        total_flops = plan.train_flops(P)
        line = {
            'metric': 'clips/sec (T=32, 480x854) fwd+bwd', 'value': round(world * args.steps / dt, 4),
            'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': 'SlowFastLayers (sp=%d, fp=%d) fwd+loss+bwd+SGD, 1 clip/GPU/step, 5 FPN levels of a '
                                   '480x854 frame (P=%d); input = %s clip resident in HBM (PackedClip hand-over, '
                                   'no layout pass in the timed region)'
                                   % (args.sp, args.fp, P, ('channel-group-major [C/32][pos][32] bf16' if grouped else
                                                            'pyramid NDHWC %s' % args.precision)),
                       'parallelism': 'dp%d' % world, 'backend': (dist.get_backend() if dist_on else 'none'),
                       'allreduce_collectives_issued': bucket.collectives,
                       'grad_accumulation': 2, 'hip_streams_per_clip': args.streams,
                       'hip_graph': ('GraphedStep: one hipGraph per clip of the accumulate-2 cycle, replayed; '
                                     'roofline.launch_ms from the per-layer pass') if args.graph else False,
                       'allreduce_buckets': ('per-layer' if args.per_layer_allreduce else 4) if dist_on else 0},
            'dropin_api_ms_per_step': None if dropin_ms is None else round(dropin_ms, 3),
            'dropin_api_note': 'same step through temporally_enhance_features([slow], [fast]) on fp32 NCHW frame lists '
                               '(reference calling convention, model.py:157-158,340): includes the fp32 NCHW -> '
                               'bf16 channel-group-major layout pass',
            'tflops_per_clip': round(total_flops / 1e12, 3),
            'achieved_tflops_whole_step': round(total_flops * world * args.steps / dt / 1e12, 2),
            'roofline': roofline,
            'kernel_events': {'timed_region': args.kernel_events,
                              'layer_table_steps': table_steps if kern is not kern_dom else None,
                              'note': 'roofline.launch_ms: HIP events around the dominant kernel inside the timed region; '
                                      'hbm_bound_passes / mfma_layers / hbm_layers / kernels_ms: a separate pass of the same '
                                      'step on ONE stream with every launch bracketed (bracketing all ~100 launches costs '
                                      '3 % of the step; concurrent kernels would stretch each other\'s durations)'},
            'hbm_bound_passes': hbm,
            'mfma_layers': mfma,
            'hbm_layers': hbm_layers,
            'kernels_ms': {k: [v[0], round(v[1], 4)] for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:28]},
        }
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(args.sp, args.fp, args.cpu_threads, args.cpu_clips)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
        sys.stdout.flush()
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
