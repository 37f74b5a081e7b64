"""micro-bench of sfvos_conv3d / wgrad on the headline shapes (whole DAVIS pyramid, bf16)"""
import ctypes
import os
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sfvos_amd import _lib, davis_pyramid  # noqa: E402

# SFVOS_LEGACY=1 + SFVOS_LIB=<round-1 build>: A/B against the round-1 ABI (no struct_size, a `zeros` argument)
LEGACY = os.environ.get('SFVOS_LEGACY') == '1'
if LEGACY:
    class LegacyDesc(ctypes.Structure):
        _fields_ = [f for f in _lib.ConvDesc._fields_ if f[0] != 'struct_size']
    _raw = ctypes.CDLL(_lib.LIB_PATH)
    _Z = None

    class _Shim(object):
        ConvDesc, BF16, FP8, MAX_LEVELS = LegacyDesc, _lib.BF16, _lib.FP8, _lib.MAX_LEVELS
        make_pyramid = staticmethod(_lib.make_pyramid)

        @staticmethod
        def load():
            return _raw

        @staticmethod
        def call(name, *args):
            global _Z
            if _Z is None:
                _Z = torch.zeros(1024, dtype=torch.uint8, device='cuda')
            if name in ('sfvos_conv3d', 'sfvos_conv3d_wgrad'):
                args = args[:-1] + (ctypes.c_void_p(_Z.data_ptr()), args[-1])
            conv = []
            for a in args:
                if isinstance(a, float):
                    a = ctypes.c_float(a)
                conv.append(a)
            fn = getattr(_raw, name)
            fn.restype = ctypes.c_int
            rc = fn(*conv)
            if rc != 0:
                raise RuntimeError('%s failed %d' % (name, rc))
    _raw.sfvos_conv3d_wgrad_workspace_bytes.restype = ctypes.c_size_t
    _lib = _Shim

P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [s for _, s in davis_pyramid()]


def desc(T, cin, cout, kt, taps, pad_t=0, shapes=SHAPES):
    d = _lib.ConvDesc()
    d.dtype, d.batch, d.t_in, d.t_alloc, d.t_offset = _lib.BF16, 1, T, T, 0
    d.c_in, d.c_out, d.kt, d.taps, d.pad_t, d.ld_x, d.ld_y, d.accumulate = cin, cout, kt, taps, pad_t, cin, cout, 0
    d.pyr = _lib.make_pyramid(shapes)
    return d, T + 2 * pad_t - kt + 1


COLD = os.environ.get('MB_COLD') == '1'   # every timed call behind a 640 MB write: operands come from HBM, not from the
_FLUSH = None                               # 256 MB Infinity Cache that back-to-back repetitions of a small layer live in


def timeit(f, reps):
    global _FLUSH
    f(); torch.cuda.synchronize()
    if COLD:
        if _FLUSH is None:
            _FLUSH = torch.empty(160 * 1024 * 1024, dtype=torch.float32, device='cuda')
        ts = []
        for _ in range(reps):
            _FLUSH.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ZERO = os.environ.get('MB_FILL') == 'zeros'   # all-zero operands: same instruction stream, lower power -> the clock the chip CAN hold


def conv(name, T, cin, cout, kt, taps, pad_t=0, reps=5, shapes=SHAPES, acc=0):
    d, t_out = desc(T, cin, cout, kt, taps, pad_t, shapes)
    d.accumulate = acc
    pix = sum(h * w for h, w in shapes)
    x = torch.randn(T * pix, cin, device='cuda').bfloat16()
    wp = (torch.randn(cout * cin * kt * taps, device='cuda') * 0.02).bfloat16()
    if ZERO:
        x.zero_(); wp.zero_()
    y = torch.empty(t_out * pix, cout, device='cuda', dtype=torch.bfloat16)
    rows = _lib.load().sfvos_conv3d_stat_rows(ctypes.byref(d), None)
    part = torch.empty(rows, 2, cout, device='cuda')
    pp = None if pad_t > 0 else P(part)
    ms = timeit(lambda: _lib.call('sfvos_conv3d', ctypes.byref(d), P(x), P(wp), None, P(y), pp, S()), reps)
    fl = 2.0 * cin * cout * kt * taps * t_out * pix
    print('conv  %-22s dbg=%s %8.3f ms %7.1f TF/s' % (name, os.environ.get('SFVOS_CONV_DEBUG', '0'), ms, fl / ms / 1e9),
          flush=True)


def conv_fp8(name, T, cin, cout, kt, reps=5, shapes=SHAPES):
    """e4m3 operands, x in 64-channel groups (fast_conv1 forward shape)"""
    d, t_out = desc(T, cin, cout, kt, 9, 0, shapes)
    d.dtype = _lib.FP8
    pix = sum(h * w for h, w in shapes)
    M = T * pix
    d.ld_x, d.x_group_stride = 64, M * 64
    x = torch.randint(0, 256, (cin // 64, M, 64), device='cuda', dtype=torch.uint8)
    x = torch.where((x & 0x7f) > 0x7d, x & 0xf0, x)            # no NaN encodings
    w = torch.randn(cout, cin, kt, 3, 3, device='cuda') * 0.02
    wp = torch.empty(w.numel(), dtype=torch.uint8, device='cuda')
    bd = torch.empty((3, cout), device='cuda')
    _lib.call('sfvos_pack_weights_fp8', P(w), None, P(wp), P(bd), cout, cin, kt, 9, 32.0, S())
    y = torch.empty(t_out * pix, cout, device='cuda', dtype=torch.bfloat16)
    rows = _lib.load().sfvos_conv3d_stat_rows(ctypes.byref(d), None)
    part = torch.empty(rows, 2, cout, device='cuda')
    ms = timeit(lambda: _lib.call('sfvos_conv3d', ctypes.byref(d), P(x), P(wp), P(bd), P(y), P(part), S()), reps)
    fl = 2.0 * cin * cout * kt * 9 * t_out * pix
    print('conv  %-22s fp8     %8.3f ms %7.1f TF/s' % (name, ms, fl / ms / 1e9), flush=True)


def wgrad(name, T, cin, cout, kt, taps, reps=5, shapes=SHAPES):
    d, t_out = desc(T, cin, cout, kt, taps, 0, shapes)
    pix = sum(h * w for h, w in shapes)
    x = torch.randn(T * pix, cin, device='cuda').bfloat16()
    dy = torch.randn(t_out * pix, cout, device='cuda').bfloat16()
    if ZERO:
        x.zero_(); dy.zero_()
    gw = torch.empty(cout * cin * kt * taps, device='cuda')
    ws = torch.empty(_lib.load().sfvos_conv3d_wgrad_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device='cuda')
    ms = timeit(lambda: _lib.call('sfvos_conv3d_wgrad', ctypes.byref(d), P(x), P(dy), P(gw), 0, P(ws), S()), reps)
    fl = 2.0 * cin * cout * kt * taps * t_out * pix
    print('wgrad %-22s       %8.3f ms %7.1f TF/s  (slab %.0f MB)' % (name, ms, fl / ms / 1e9, ws.numel() / 1e6), flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else 'all'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if which in ('all', 'f1'):
    conv('f1 256->32 k11', 32, 256, 32, 11, 9, reps=reps)
if which == 'f1f8':
    conv_fp8('f1 256->32 k11', 32, 256, 32, 11, reps=reps)
    conv('f1 256->32 k11', 32, 256, 32, 11, 9, reps=reps)
if which in ('all', 'wf1'):
    wgrad('f1 256->32 k11', 32, 256, 32, 11, 9, reps=reps)
if which == 'latf':
    conv('l1 32->64 k20', 22, 32, 64, 20, 1, reps=reps)
    conv('l2 32->64 k11', 12, 32, 64, 11, 1, reps=reps)
if which == 'lat':
    conv('dgrad l1 64->32 k20', 3, 64, 32, 20, 1, pad_t=19, reps=reps, acc=1)
    conv('dgrad l2 64->32 k11', 2, 64, 32, 11, 1, pad_t=10, reps=reps, acc=1)
if which == 'wide':
    conv('s1 256->192 k2', 4, 256, 192, 2, 9, reps=reps)
    conv('s2 256->192 k2', 3, 256, 192, 2, 9, reps=reps)
    conv('s3 256->224 k2', 2, 256, 224, 2, 9, reps=reps)
    conv('dgrad s2 192->256', 2, 192, 256, 2, 9, pad_t=1, reps=reps)
    conv('dgrad s3 224->256', 1, 224, 256, 2, 9, pad_t=1, reps=reps)
if which == 'wall':
    wgrad('f1 256->32 k11', 32, 256, 32, 11, 9, reps=reps)
    wgrad('s1 256->192 k2', 4, 256, 192, 2, 9, reps=reps)
    wgrad('s2 256->192 k2', 3, 256, 192, 2, 9, reps=reps)
    wgrad('s3 256->224 k2', 2, 256, 224, 2, 9, reps=reps)
    wgrad('f2 32->32 k11', 22, 32, 32, 11, 9, reps=reps)
    wgrad('f3 32->32 k12', 12, 32, 32, 12, 9, reps=reps)
    wgrad('l1 32->64 k20', 22, 32, 64, 20, 1, reps=reps)
    wgrad('l2 32->64 k11', 12, 32, 64, 11, 1, reps=reps)
if which == 'wlat':   # the HBM-bound weight gradients: laterals and fast_conv3 of (4,32) and (4,64)
    wgrad('l1 32->64 k20', 22, 32, 64, 20, 1, reps=reps)
    wgrad('l2 32->64 k11', 12, 32, 64, 11, 1, reps=reps)
    wgrad('f3 32->32 k12', 12, 32, 32, 12, 9, reps=reps)
    wgrad('C4 l1 32->64 k41', 43, 32, 64, 41, 1, reps=reps)
    wgrad('C4 l2 32->64 k21', 22, 32, 64, 21, 1, reps=reps)
    wgrad('C4 f3 32->32 k22', 22, 32, 32, 22, 9, reps=reps)
if which == 'f3':
    conv('f3 32->32 k12', 12, 32, 32, 12, 9, reps=reps)
    conv('dgrad f3 32->32 k12', 1, 32, 32, 12, 9, pad_t=11, reps=reps, acc=1)
if which == 'f2':
    conv('f2 32->32 k11', 22, 32, 32, 11, 9, reps=reps)
if which == 'df2':
    conv('dgrad f2 32->32 k11', 12, 32, 32, 11, 9, pad_t=10, reps=reps)
if which == 'all':
    conv('s1 256->192 k2', 4, 256, 192, 2, 9)
    conv('s3 256->224 k2', 2, 256, 224, 2, 9)
    conv('f2 32->32 k11', 22, 32, 32, 11, 9)
    conv('l1 32->64 k20 1x1', 22, 32, 64, 20, 1)
    conv('dgrad s2 192->256', 2, 192, 256, 2, 9, pad_t=1)
    conv('dgrad l1 64->32 k20', 3, 64, 32, 20, 1, pad_t=19)
    conv('dgrad f2 32->32 k11', 12, 32, 32, 11, 9, pad_t=10)
    wgrad('s1 256->192 k2', 4, 256, 192, 2, 9)
    wgrad('f2 32->32 k11', 22, 32, 32, 11, 9)
    wgrad('l1 32->64 k20', 22, 32, 64, 20, 1)
