#!/usr/bin/env python
"""Sequence-mode inference throughput of the SlowFastLayers path: frames/s of
sfvos_amd.SlowFastStream (one new frame of every activation stream per video frame) next to the per-frame
recompute the reference's evaluation loop does (the whole fp-frame window through the module for every frame,
code/helpers/model.py:316-340), same kernels, same box, eval mode, DAVIS pyramid of a 480x854 frame, bf16.

    python bench_stream.py [--sp 4 --fp 32] [--frames 40] [--precision bf16]

Prints ONE JSON line.  Not the headline metric (bench.py is): an extra measurement for the serving use of the path."""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sp', type=int, default=4)
    ap.add_argument('--fp', type=int, default=32)
    ap.add_argument('--frames', type=int, default=40, help='timed frames (after the pipeline is full)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--chunk', type=int, default=4, help='frames per push_many() call of the chunked run')
    args = ap.parse_args()
    from sfvos_amd import PackedClip, SlowFastLayers, SlowFastStream, davis_pyramid
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(63)
    model = SlowFastLayers(256, dev, args.sp, args.fp, precision=args.precision).to(dev).eval()
    pyr = davis_pyramid()
    keys = [k for k, _ in pyr]
    shapes = [s for _, s in pyr]
    gen = torch.Generator(device=dev).manual_seed(7)
    pool = [OrderedDict((k, torch.randn((256, h, w), generator=gen, device=dev)) for k, (h, w) in pyr)
            for _ in range(4)]   # a few distinct frames, cycled

    res = {}
    with torch.no_grad():
        for chunk in (1, args.chunk):
            stream = SlowFastStream(model, shapes, keys=keys, chunk=chunk)
            fill = ((args.fp + 2 + chunk - 1) // chunk) * chunk
            for i in range(0, fill, chunk):          # fill the pipeline (+ warm-up)
                stream.push_many([pool[(i + j) % len(pool)] for j in range(chunk)])
            torch.cuda.synchronize()
            nfr = (args.frames // chunk) * chunk
            t0 = time.time()
            for i in range(0, nfr, chunk):
                out = stream.push_many([pool[(i + j) % len(pool)] for j in range(chunk)])
            torch.cuda.synchronize()
            res[chunk] = (time.time() - t0) / nfr
            assert len(out) == chunk
            del stream
        dt_stream = res[1]
        # per-frame recompute: the whole window through the module (channels-last hand-over, forward only)
        tdt = torch.bfloat16 if args.precision == 'bf16' else torch.float32
        levels = [torch.randn((1, args.fp, h, w, 256), generator=gen, device=dev).to(tdt) for _, (h, w) in pyr]
        clip = PackedClip.from_levels(levels, keys=keys, layout='grouped' if args.precision == 'bf16' else 'ndhwc')
        del levels
        for _ in range(3):
            model.enhance_packed(clip)
        torch.cuda.synchronize()
        reps = max(5, args.frames // 4)
        t0 = time.time()
        for _ in range(reps):
            model.enhance_packed(clip)
        torch.cuda.synchronize()
        dt_full = (time.time() - t0) / reps
        # the whole window from fp32 frames (the reference's calling convention), bf16 vs the e4m3 fast_conv1 path
        win = {}
        frames_fast = [OrderedDict((k, torch.randn((args.fp, 256, h, w), generator=gen, device=dev)) for k, (h, w) in pyr)]
        c = args.fp // 2
        frames_slow = [OrderedDict((k, v[c - args.sp // 2: c + (args.sp + 1) // 2]) for k, v in frames_fast[0].items())]
        for prec in ('bf16', 'fp8'):
            m2 = SlowFastLayers(256, dev, args.sp, args.fp, precision=prec).to(dev).eval()
            m2.load_state_dict(model.state_dict())
            for _ in range(2):
                m2.temporally_enhance_features(frames_slow, frames_fast)
            timer = m2.enable_kernel_timer()
            timer.reset()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(5):
                m2.temporally_enhance_features(frames_slow, frames_fast)
            torch.cuda.synchronize()
            win[prec] = {'ms': round((time.time() - t0) / 5 * 1e3, 3),
                         'fast_conv1_ms': round(timer.summary()['conv_fwd/f1'][1], 3)}
            del m2
    P = sum(h * w for h, w in shapes)
    plan = model.plan
    flops_stream = sum(2.0 * l.c_in * l.c_out * l.kt * l.taps * P for l in plan.layers)   # one output frame per layer
    print(json.dumps({
        'metric': 'frames/sec, sequence inference (T=%d window, 480x854)' % args.fp, 'unit': 'frames/s',
        'stream': {'value': round(1.0 / dt_stream, 2), 'ms_per_frame': round(1e3 * dt_stream, 3),
                   'gflop_per_frame': round(flops_stream / 1e9, 1)},
        'stream_chunked': {'chunk': args.chunk, 'value': round(1.0 / res[args.chunk], 2),
                           'ms_per_frame': round(1e3 * res[args.chunk], 3)},
        'recompute_window_per_frame': {'value': round(1.0 / dt_full, 2), 'ms_per_frame': round(1e3 * dt_full, 3),
                                       'gflop_per_frame': round(plan.forward_flops(P) / 1e9, 1),
                                       'note': 'window already channels-last on the GPU; the reference also restacks it'},
        'window_from_fp32_frames': win,
        'speedup': round(dt_full / dt_stream, 2), 'speedup_chunked': round(dt_full / res[args.chunk], 2), 'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': 'SlowFastLayers (sp=%d, fp=%d) eval forward per video frame, 5 FPN levels (P=%d)'
                               % (args.sp, args.fp, P)}}))


if __name__ == '__main__':
    main()
