"""Randomised shapes for the bf16 forward convs and data gradients (frame-split family incl. the one-frame and the
input-stationary kernels, wide kernel, laterals) against torch on the CPU.
usage: python tools/diag/fuzz_conv.py [cases] [seed]"""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
import torch
import torch.nn.functional as F
from sfvos_amd import _lib
from test_gpu_kernels import P, S, from_pyr, make_desc, relmax, to_pyr

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
_lib.load()
worst = 0.0
for case in range(n_cases):
    kind = rng.choice(['fs', 'fs_t1', 'wide', 'lat'])
    B = rng.choice([1, 1, 2])
    shapes = [(rng.randint(1, 40), rng.randint(1, 70)) for _ in range(rng.choice([1, 2, 3]))]
    if kind == 'fs':
        cin, cout, taps, kt, t_out = rng.choice([32, 64, 256]), 32, 9, rng.randint(1, 6), rng.randint(1, 7)
    elif kind == 'fs_t1':
        cin, cout, taps, kt, t_out = 32, 32, 9, rng.randint(2, 12), 1
    elif kind == 'wide':
        cin, cout, taps, kt, t_out = rng.choice([64, 256]), rng.choice([128, 192, 224, 256]), 9, rng.randint(1, 3), rng.randint(1, 4)
        shapes = shapes[:2]
    else:
        cin, cout, taps, kt, t_out = 32, 64, 1, rng.randint(1, 24), rng.randint(1, 3)
    T = kt + t_out - 1
    g = torch.Generator().manual_seed(2000 + case)
    k = 3 if taps == 9 else 1
    w = (torch.randn(cout, cin, kt, k, k, generator=g) / np.sqrt(cin * kt * taps)).bfloat16().float()
    bias = torch.randn(cout, generator=g)
    xs = [torch.randn(B, cin, T, H, W, generator=g).bfloat16().float().requires_grad_(True) for (H, W) in shapes]
    dys = [torch.randn(B, cout, t_out, H, W, generator=g).bfloat16().float() for (H, W) in shapes]
    refs = []
    for x, dy in zip(xs, dys):
        y = F.conv3d(x, w, bias, padding=(0, 1, 1) if taps == 9 else 0)
        y.backward(dy)
        refs.append(y.detach())
    # forward
    xd = to_pyr([x.detach() for x in xs], 'bf16')
    d, t_o = make_desc(_lib, 'bf16', B, T, shapes, cin, cout, kt, taps, 0, cin, cout)
    wp = torch.empty(w.numel(), dtype=torch.bfloat16, device='cuda')
    _lib.call('sfvos_pack_weights_fwd', P(w.cuda()), P(wp), d.dtype, cout, cin, kt, taps, S())
    y = torch.empty((sum(B * t_out * H * W for H, W in shapes), cout), dtype=torch.bfloat16, device='cuda')
    _lib.call('sfvos_conv3d', ctypes.byref(d), P(xd), P(wp), P(bias.cuda()), P(y), None, S())
    ef = max(relmax(a, b) for a, b in zip(from_pyr(y, B, cout, t_out, shapes), refs))
    # data gradient
    dyd = to_pyr(dys, 'bf16')
    dd, t_back = make_desc(_lib, 'bf16', B, t_out, shapes, cout, cin, kt, taps, kt - 1, cout, cin)
    wd = torch.empty(w.numel(), dtype=torch.bfloat16, device='cuda')
    _lib.call('sfvos_pack_weights_dgrad', P(w.cuda()), P(wd), dd.dtype, cout, cin, kt, taps, S())
    dx = torch.empty((sum(B * T * H * W for H, W in shapes), cin), dtype=torch.bfloat16, device='cuda')
    _lib.call('sfvos_conv3d', ctypes.byref(dd), P(dyd), P(wd), None, P(dx), None, S())
    eb = max(relmax(a, x.grad) for a, x in zip(from_pyr(dx, B, cin, T, shapes), xs))
    worst = max(worst, ef, eb)
    flag = '' if max(ef, eb) < 2e-2 else '   <-- FAIL'
    print('%-5s B %d shapes %-30s %3d->%3d kt %2d t_out %d: fwd %.2e dgrad %.2e%s'
          % (kind, B, shapes, cin, cout, kt, t_out, ef, eb, flag), flush=True)
    if flag:
        sys.exit(1)
print('worst %.2e over %d cases' % (worst, n_cases))
