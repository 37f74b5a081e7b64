"""Randomised shapes for the bf16 weight-gradient kernels of the 32-channel 3x3 layers (wgrad_t1.hip: one output frame;
wgrad_kernel's row-split configuration: several), of the laterals, and of the c_in >= 64 3x3 layers ('wide': frame-paired
stages when t_out is even and >= 4, windows that may reach outside the x buffer on either side), against torch on the CPU.
usage: python tools/diag/fuzz_wgrad.py [cases] [seed]"""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
import torch.nn.functional as F
from sfvos_amd import _lib
from test_gpu_kernels import P, S, make_desc, relmax, to_pyr

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
_lib.load()
worst = 0.0
for case in range(n_cases):
    kind = rng.choice(['t1', 'ks', 'lat', 'wide', 'wide'])
    B = rng.choice([1, 1, 2])
    L = rng.choice([1, 2, 3])
    shapes = [(rng.randint(1, 70), rng.randint(1, 70)) for _ in range(L)]
    if kind == 't1':
        cin, cout, taps, kt = 32, 32, 9, rng.randint(2, 14); t_out = 1
    elif kind == 'ks':
        cin, cout, taps, kt = 32, 32, 9, rng.randint(1, 12); t_out = rng.randint(2, 5)
    elif kind == 'wide':
        cin, cout = rng.choice([64, 96, 128, 256]), rng.choice([32, 32, 64, 192, 224])
        taps, kt, t_out = 9, rng.randint(1, 12), rng.choice([4, 4, 6, 8, 10, 2, 3, 5])
        shapes = [(rng.randint(1, 40), rng.randint(1, 40)) for _ in range(L)]
    else:
        cin, cout, taps, kt = 32, 64, 1, rng.randint(1, 24); t_out = rng.randint(1, 3)
    T = kt + t_out - 1
    off = rng.randint(0, 2)
    Ta = T + off + rng.randint(0, 2)
    if kind == 'wide' and rng.random() < 0.5:   # the window reaches outside the buffer: those frames are zeros
        off = rng.randint(-2, 2)
        Ta = max(1, T + off - rng.randint(0, 2))
    ld_x, ld_y = cin + 8 * rng.randint(0, 2), cout + 8 * rng.randint(0, 4)
    g = torch.Generator().manual_seed(1000 + case)
    k = 3 if taps == 9 else 1
    w = torch.zeros(cout, cin, kt, k, k, requires_grad=True)
    xs = [torch.randn(B, cin, Ta, H, W, generator=g).bfloat16().float() for (H, W) in shapes]
    dys = [torch.randn(B, cout, t_out, H, W, generator=g).bfloat16().float() for (H, W) in shapes]
    for x, dy in zip(xs, dys):
        xw = torch.zeros(B, cin, T, x.shape[3], x.shape[4])
        lo, hi = max(0, off), min(Ta, off + T)
        if hi > lo:
            xw[:, :, lo - off:hi - off] = x[:, :, lo:hi]
        F.conv3d(xw, w, None, padding=(0, 1, 1) if taps == 9 else 0).backward(dy)
    ref = w.grad
    xd, dyd = to_pyr(xs, 'bf16', ld_x), to_pyr(dys, 'bf16', ld_y)
    d, _ = make_desc(_lib, 'bf16', B, T, shapes, cin, cout, kt, taps, 0, ld_x, ld_y, t_alloc=Ta, t_offset=off)
    nbytes = _lib.load().sfvos_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
    ws = torch.full((nbytes // 4,), float('nan'), dtype=torch.float32, device='cuda')
    gw = torch.full(ref.shape, 5.0, dtype=torch.float32, device='cuda')
    _lib.call('sfvos_conv3d_wgrad', ctypes.byref(d), P(xd), P(dyd), P(gw), 0, P(ws), S())
    e = relmax(gw.cpu(), ref)
    worst = max(worst, e)
    flag = '' if e < 2e-2 else '   <-- FAIL'
    print('%-4s B %d shapes %-28s kt %2d t_out %d window [%d,+%d) of %d ld %d/%d: %.2e%s'
          % (kind, B, shapes, kt, t_out, off, T, Ta, ld_x, ld_y, e, flag), flush=True)
    if flag:
        sys.exit(1)
print('worst %.2e over %d cases' % (worst, n_cases))
