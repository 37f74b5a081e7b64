"""bench.py -- clips/s of the SlowFastLayers hot path (fwd + bwd + SGD) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[1] restated in the reference's own parametrisation:
(slow, fast) = (4, 32) frames, one clip = one call of temporally_enhance_features over the five FPN
levels of a 480x854 DAVIS frame (P = 85 932 positions), bf16 activations / f32 accumulate, inputs
resident in HBM as channels-last clips, synthetic N(0,1) features, random-init weights.
A step = forward + backward of one clip; gradients are accumulated over 2 clips and then the
optimiser steps (reference model.py:369-374), with the data-parallel all-reduce in front of it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--sp', type=int, default=4)
    ap.add_argument('--fp', type=int, default=32)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--streams', type=int, default=1,
                    help='HIP streams per clip: 1 = every launch on one stream (clean per-kernel HIP-event / rocprof '
                         'durations, the default here); 2 = slow pathway on a side stream (module default, ~5 %% '
                         'more clips/s, but concurrent kernels stretch each other\'s measured duration)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--ndhwc-input', action='store_true', help='hand the clip over as plain pyramid NDHWC (A/B of the grouped layout)')
    ap.add_argument('--no-grad-sink', action='store_true',
                    help='let autograd accumulate parameter gradients from temporaries (A/B of FusedSGD.attach)')
    ap.add_argument('--cpu-threads', type=int, default=0)
    return ap.parse_args()


def cpu_baseline(sp, fp, threads):
    """The oracle (torch-CPU restatement of the reference path) timed on this box's host cores on a
    bounded sample: ONE clip's FPN level '0' (192x336 = 75 % of the clip's positions, 10-20 s), fwd+bwd."""
    from oracle.slowfast_ref import OracleSlowFastLayers, proxy_loss
    from sfvos_amd import davis_pyramid
    if threads > 0:
        torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    pyr = dict(davis_pyramid())
    level = '0'
    H, W = pyr[level]
    P = sum(h * w for h, w in pyr.values())
    torch.manual_seed(63)
    m = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    m.train()
    fast = torch.randn(1, 256, fp, H, W)
    slow = fast[:, :, fp // 2 - sp // 2: fp // 2 + (sp + 1) // 2]
    t0 = time.time()
    s, f = m(slow, fast)
    loss = proxy_loss({level: torch.cat([s, f], 1).squeeze(2)})
    loss.backward()
    dt = time.time() - t0
    frac = float(H * W) / P
    return {'value': frac / dt, 'unit': 'clips/s', 'cores': cores, 'kind': 'port',
            'sample': "oracle fwd+bwd of FPN level '%s' (%dx%d, %.1f%% of one clip's positions) in %.1f s, "
                      "scaled by position share" % (level, H, W, 100 * frac, dt)}


def pmc_traffic(path):
    """HBM bytes of the dominant kernel per launch, from the committed rocprofv3 PMC passes of this same
    command (FETCH_SIZE and WRITE_SIZE in separate passes; KB units; gfx950 FETCH_SIZE counts half of a wide
    streaming read, so it is doubled -- MI355X_MICROARCH.md, HBM).  None when the summary is absent."""
    try:
        with open(path) as f:
            s = json.load(f)
        k = 'conv3d_fs_kernel<1, 256>'
        return 2.0 * s[k]['FETCH_SIZE']['mean'] * 1024 + s[k]['WRITE_SIZE']['mean'] * 1024
    except Exception:
        return None


def main():
    args = parse()
    from sfvos_amd import FusedSGD, GradBucket, PackedClip, SlowFastLayers, davis_pyramid, init_distributed
    from oracle.slowfast_ref import proxy_loss  # loss definition only (SURVEY.md 8d stand-in for RoI-head losses)

    rank, world, local = init_distributed()
    if world != args.gpus:
        if rank == 0 and world > 1:
            print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)
    dev = torch.device('cuda', (local % torch.cuda.device_count()) if world > 1 else 0)
    torch.cuda.set_device(dev)

    torch.manual_seed(63)
    model = SlowFastLayers(256, dev, args.sp, args.fp, precision=args.precision).to(dev)
    model.train()
    model.n_streams = args.streams
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    bucket = GradBucket(opt.flat_grad)
    if not args.no_grad_sink:
        # kernels write parameter gradients straight into the flat gradient buffer; with world > 1 each layer's
        # slice is all-reduced (RCCL, side stream) as soon as backward has produced it
        opt.attach(model, bucket)
    tdt = torch.bfloat16 if args.precision == 'bf16' else torch.float32
    pyr = davis_pyramid()
    P = sum(h * w for _, (h, w) in pyr)
    gen = torch.Generator(device=dev).manual_seed(63 + rank)
    # one clip = the fast window of every FPN level, channels-last, already in the pyramid layout
    levels = [torch.randn((1, args.fp, h, w, 256), generator=gen, device=dev, dtype=torch.float32).to(tdt)
              for _, (h, w) in pyr]
    # bf16: the clip is handed over channel-group-major (64-byte groups), the layout the first convs read fastest
    clip = PackedClip.from_levels(levels, keys=[k for k, _ in pyr],
                                  layout='grouped' if (args.precision == 'bf16' and not args.ndhwc_input) else 'ndhwc')
    del levels

    timer = model.enable_kernel_timer()

    def step(i):
        out = model.enhance_packed(clip)
        loss = proxy_loss(out)
        if i % 2 == 1:  # model.py:372-374: optimiser every 2nd clip: this backward completes the gradients
            bucket.arm()
        loss.backward()
        if i % 2 == 1:
            bucket.finish()
            opt.step()
            opt.zero_grad()

    for i in range(args.warmup):
        step(i)
    timer.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        plan = model.plan
        kern = timer.summary()  # name -> (calls, mean ms)
        # dominant kernel: fast_conv1 forward over the whole 5-level pyramid (72 % of the forward
        # FLOPs).  It runs as two back-to-back launches of conv3d_kernel<bf16,9,9,TT,1,1,8,1,256>: output frames
        # 0-15 in blocks of TT=4 and frames 16-21 in blocks of TT=3 (no padded frame); the HIP events bracket both.
        l = plan.layer('f1')
        dom_flops = 2.0 * l.c_in * l.c_out * l.kt * l.taps * l.t_out * P
        dom = kern.get('conv_fwd/f1')
        peak = 2500.0 if args.precision == 'bf16' else 157.3
        roofline = None
        if dom:
            ach = dom_flops / (dom[1] * 1e-3) / 1e12
            roofline = {'bound': 'mfma',
                        'kernel': 'sfvos::conv3d_fs_kernel<1,256> (fast_conv1 forward, 256->32 ch, %dx3x3, '
                                  '%d->%d frames, 5-level pyramid, one launch)' % (l.kt, l.t_in, l.t_out),
                        'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                        'launch_ms': round(dom[1], 4), 'flops_per_launch': dom_flops,
                        'traffic': pmc_traffic(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json'))
                        if (args.sp, args.fp, args.precision) == (4, 32, 'bf16') else None}
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
                    mfma[k] = {'ms': round(kern[k][1], 4), 'TFLOPs': round(fl[l.name] / kern[k][1] / 1e9, 1)}
        total_flops = plan.train_flops(P)
        line = {
            'metric': 'clips/sec (T=32, 480x854) fwd+bwd', 'value': round(world * args.steps / dt, 4),
            'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': 'SlowFastLayers (sp=%d, fp=%d) fwd+bwd+SGD, 1 clip/GPU/step, 5 FPN levels of a '
                                   '480x854 frame (P=%d), NDHWC inputs resident in HBM' % (args.sp, args.fp, P),
                       'parallelism': 'dp%d' % world, 'grad_accumulation': 2, 'hip_streams_per_clip': args.streams},
            'tflops_per_clip': round(total_flops / 1e12, 3),
            'achieved_tflops_whole_step': round(total_flops * world * args.steps / dt / 1e12, 2),
            'roofline': roofline,
            'hbm_bound_passes': hbm,
            'mfma_layers': mfma,
            'kernels_ms': {k: [v[0], round(v[1], 4)] for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:24]},
        }
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(args.sp, args.fp, args.cpu_threads)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
