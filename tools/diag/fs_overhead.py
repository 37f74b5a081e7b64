"""Fixed cost per workgroup of the frame-split conv kernel: a 32 -> 32 (or 256 -> 32) kt x 3 x 3 conv with 12 output frames
over the DAVIS pyramid, timed for several kt: time = rounds * (a + b * stages); the intercept a is what a workgroup spends
outside its K loop (launch, setup, first loads, epilogue).  usage: python tools/diag/fs_overhead.py [cin] [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sfvos_amd import _lib, davis_pyramid
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [s for _, s in davis_pyramid()]
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t_out = 12
pix = sum(h * w for h, w in SHAPES)
pts = []
for kt in (1, 2, 4, 6, 8, 11, 16, 22, 33):
    T = kt + t_out - 1
    d = _lib.ConvDesc()
    d.dtype, d.batch, d.t_in, d.t_alloc, d.t_offset = _lib.BF16, 1, T, T, 0
    d.c_in, d.c_out, d.kt, d.taps, d.pad_t, d.ld_x, d.ld_y, d.accumulate = cin, 32, kt, 9, 0, cin, 32, 0
    d.pyr = _lib.make_pyramid(SHAPES)
    x = torch.randn(T * pix, cin, device='cuda').bfloat16()
    wp = (torch.randn(32 * cin * kt * 9, device='cuda') * 0.02).bfloat16()
    y = torch.empty(t_out * pix, 32, device='cuda', dtype=torch.bfloat16)
    rows = _lib.load().sfvos_conv3d_stat_rows(ctypes.byref(d), None)
    part = torch.empty(rows, 2, 32, device='cuda')
    f = lambda: _lib.call('sfvos_conv3d', ctypes.byref(d), P(x), P(wp), None, P(y), P(part), S())
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    stages = kt * (cin // 32)
    fl = 2.0 * cin * 32 * kt * 9 * t_out * pix
    pts.append((stages, ms))
    print('kt %2d  stages/WG %3d  %8.3f ms  %7.1f TF/s (%.3f of 2.5 PF)' % (kt, stages, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500), flush=True)
    del x, wp, y
# least squares over the points with >= 4 stages
import numpy as np
a = np.array([(s, 1.0) for s, _ in pts if s >= 4]); b = np.array([m for s, m in pts if s >= 4])
slope, icpt = np.linalg.lstsq(a, b, rcond=None)[0]
print('time = %.4f ms + %.5f ms per stage: the fixed part equals %.1f stages' % (icpt, slope, icpt / slope))
