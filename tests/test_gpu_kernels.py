"""GPU: every libsfvos kernel, called through the C ABI (ctypes), against a plain PyTorch fp32
reference of the same op computed on the CPU.  Tolerances: fp32 path 1e-4 relative to the
tensor scale (exact-f32 MFMA, only summation order differs); bf16 path 2e-2 (bf16 operands,
f32 accumulate) -- stated per assert."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {'fp32': 1e-4, 'bf16': 2e-2}
TDT = {'fp32': torch.float32, 'bf16': torch.bfloat16}


@pytest.fixture(scope='module')
def lib():
    from sfvos_amd import _lib
    lib = _lib.load()
    _lib.check(lib.sfvos_check_device(), 'sfvos_check_device')
    return _lib


def P(t, off=0):
    return ctypes.c_void_p(t.data_ptr() + off * t.element_size())


def S():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def relmax(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def make_desc(_lib, prec, B, T, H, W, cin, cout, kt, taps, pad_t, ld_x, ld_y, acc=0):
    d = _lib.ConvDesc()
    d.dtype = _lib.F32 if prec == 'fp32' else _lib.BF16
    d.batch, d.t_in, d.h, d.w, d.c_in, d.c_out, d.kt, d.taps, d.pad_t = B, T, H, W, cin, cout, kt, taps, pad_t
    d.ld_x, d.ld_y, d.accumulate = ld_x, ld_y, acc
    t_out = T + 2 * pad_t - kt + 1
    d.x_batch_stride, d.y_batch_stride = T * H * W * ld_x, t_out * H * W * ld_y
    return d, t_out


def ndhwc(x_ncdhw, prec, ld=None):
    """[B,C,T,H,W] fp32 cpu -> [B,T,H,W,ld] device tensor of the compute dtype (extra channels = junk 7.0)."""
    B, C, T, H, W = x_ncdhw.shape
    ld = ld or C
    out = torch.full((B, T, H, W, ld), 7.0, dtype=torch.float32)
    out[..., :C] = x_ncdhw.permute(0, 2, 3, 4, 1)
    return out.to(TDT[prec]).cuda()


CONV_CASES = [
    # B  T   H   W  cin cout kt taps  (covers narrow / mid / wide families, ragged tiles, TT blocks)
    (1, 4, 12, 21, 256, 192, 2, 9),
    (1, 3, 9, 17, 256, 224, 2, 9),
    (2, 7, 6, 10, 64, 32, 3, 9),
    (1, 13, 20, 19, 32, 32, 2, 9),     # t_out 12 -> two frame blocks of 6
    (1, 14, 5, 33, 32, 32, 4, 9),      # t_out 11
    (1, 9, 18, 16, 32, 64, 5, 1),      # lateral, t_out 5 (TT 6)
    (1, 5, 7, 40, 32, 64, 3, 1),
    (1, 3, 10, 12, 64, 32, 2, 1),      # lateral dgrad shape (narrow, 1x1)
]


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv3d_forward_bias_stats(lib, prec, case):
    B, T, H, W, cin, cout, kt, taps = case
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, cin, T, H, W, generator=g)
    k = 3 if taps == 9 else 1
    w = torch.randn(cout, cin, kt, k, k, generator=g) / np.sqrt(cin * kt * taps)
    bias = torch.randn(cout, generator=g) * 0.1
    if prec == 'bf16':  # the reference sees the same rounded operands
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv3d(x, w, bias, padding=(0, 1, 1) if taps == 9 else 0)
    ld_x, ld_y = cin + 32, cout + 64
    xd = ndhwc(x, prec, ld_x)
    d, t_out = make_desc(lib, prec, B, T, H, W, cin, cout, kt, taps, 0, ld_x, ld_y)
    wd = w.cuda()
    wp = torch.empty(w.numel(), dtype=TDT[prec], device='cuda')
    lib.call('sfvos_pack_weights_fwd', P(wd), P(wp), d.dtype, cout, cin, kt, taps, S())
    y = torch.full((B, t_out, H, W, ld_y), -3.0, dtype=TDT[prec], device='cuda')
    rows = lib.load().sfvos_conv3d_stat_rows(ctypes.byref(d))
    assert rows > 0
    part = torch.full((rows, 2, cout), 1e9, dtype=torch.float32, device='cuda')
    zeros = torch.zeros(1024, dtype=torch.uint8, device='cuda')
    lib.call('sfvos_conv3d', ctypes.byref(d), P(xd), P(wp), P(bias.cuda()), P(y), P(part), P(zeros), S())
    torch.cuda.synchronize()
    got = y[..., :cout].float().cpu().permute(0, 4, 1, 2, 3)
    assert relmax(got, ref) < TOL[prec]
    assert torch.all(y[..., cout:].float() == -3.0), 'wrote outside its channel slice'
    s = part.double().sum(0).cpu()
    ref_s1 = ref.double().sum((0, 2, 3, 4))
    ref_s2 = (ref.double() ** 2).sum((0, 2, 3, 4))
    n = B * t_out * H * W
    assert float((s[0] - ref_s1).abs().max()) / n < 1e-3 * float(ref.abs().max())
    assert relmax(s[1], ref_s2) < (1e-4 if prec == 'fp32' else 2e-2)


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', [(1, 4, 12, 21, 256, 192, 2, 9), (1, 13, 10, 19, 32, 32, 2, 9),
                                  (1, 9, 9, 16, 32, 64, 5, 1), (1, 7, 6, 10, 256, 32, 3, 9)])
def test_conv3d_dgrad_and_accumulate(lib, prec, case):
    B, T, H, W, cin, cout, kt, taps = case
    g = torch.Generator().manual_seed(99)
    k = 3 if taps == 9 else 1
    x = torch.randn(B, cin, T, H, W, generator=g, requires_grad=True)
    w = torch.randn(cout, cin, kt, k, k, generator=g) / np.sqrt(cout * kt * taps)
    t_out = T - kt + 1
    dy = torch.randn(B, cout, t_out, H, W, generator=g)
    if prec == 'bf16':
        w, dy = w.bfloat16().float(), dy.bfloat16().float()
    F.conv3d(x, w, None, padding=(0, 1, 1) if taps == 9 else 0).backward(dy)
    ref = x.grad
    dyd = ndhwc(dy, prec)
    # the data-gradient conv: channels swapped, pad_t = kt-1
    d, t_back = make_desc(lib, prec, B, t_out, H, W, cout, cin, kt, taps, kt - 1, cout, cin)
    assert t_back == T
    wp = torch.empty(w.numel(), dtype=TDT[prec], device='cuda')
    lib.call('sfvos_pack_weights_dgrad', P(w.cuda()), P(wp), d.dtype, cout, cin, kt, taps, S())
    dx = torch.empty((B, T, H, W, cin), dtype=TDT[prec], device='cuda')
    zeros = torch.zeros(1024, dtype=torch.uint8, device='cuda')
    lib.call('sfvos_conv3d', ctypes.byref(d), P(dyd), P(wp), None, P(dx), None, P(zeros), S())
    got = dx.float().cpu().permute(0, 4, 1, 2, 3)
    assert relmax(got, ref) < TOL[prec]
    d.accumulate = 1  # second call adds: dx == 2 * grad
    lib.call('sfvos_conv3d', ctypes.byref(d), P(dyd), P(wp), None, P(dx), None, P(zeros), S())
    got2 = dx.float().cpu().permute(0, 4, 1, 2, 3)
    assert relmax(got2, 2 * ref) < 2 * TOL[prec]


WGRAD_CASES = [
    (1, 4, 12, 21, 256, 32, 2, 9),    # cfg A (c_out 32, c_in 256)
    (1, 3, 9, 17, 256, 192, 2, 9),    # cfg B
    (1, 3, 6, 10, 256, 224, 2, 9),    # cfg B, ragged n blocks
    (2, 13, 7, 19, 32, 32, 11, 9),    # cfg C, kt 11 -> two dt groups
    (1, 9, 18, 16, 32, 64, 5, 1),     # cfg D (lateral)
    (1, 24, 5, 33, 32, 64, 20, 1),
]


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', WGRAD_CASES)
def test_conv3d_wgrad(lib, prec, case):
    B, T, H, W, cin, cout, kt, taps = case
    g = torch.Generator().manual_seed(7)
    k = 3 if taps == 9 else 1
    x = torch.randn(B, cin, T, H, W, generator=g)
    w = torch.zeros(cout, cin, kt, k, k, requires_grad=True)
    t_out = T - kt + 1
    dy = torch.randn(B, cout, t_out, H, W, generator=g)
    if prec == 'bf16':
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    F.conv3d(x, w, None, padding=(0, 1, 1) if taps == 9 else 0).backward(dy)
    ref = w.grad
    ld_x, ld_y = cin + 32, cout + 32
    xd, dyd = ndhwc(x, prec, ld_x), ndhwc(dy, prec, ld_y)
    d, _ = make_desc(lib, prec, B, T, H, W, cin, cout, kt, taps, 0, ld_x, ld_y)
    nbytes = lib.load().sfvos_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    gw = torch.full(ref.shape, 5.0, dtype=torch.float32, device='cuda')
    zeros = torch.zeros(1024, dtype=torch.uint8, device='cuda')
    lib.call('sfvos_conv3d_wgrad', ctypes.byref(d), P(xd), P(dyd), P(gw), 0, P(ws), P(zeros), S())
    assert relmax(gw.cpu(), ref) < TOL[prec]
    lib.call('sfvos_conv3d_wgrad', ctypes.byref(d), P(xd), P(dyd), P(gw), 1, P(ws), P(zeros), S())
    assert relmax(gw.cpu(), 2 * ref) < 2 * TOL[prec]


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_layout_roundtrip_and_strided_source(lib, prec):
    T, C, H, W = 5, 256, 7, 13
    g = torch.Generator().manual_seed(3)
    frames = torch.randn(T, C, H, W, generator=g)
    # the reference hands over stack(frames).transpose(1, 2): a non-contiguous [C,T,H,W] view
    view = frames.cuda().unsqueeze(0).transpose(1, 2)[0]          # [C,T,H,W], strides of the frame stack
    dt = lib.F32 if prec == 'fp32' else lib.BF16
    dst = torch.empty((T, H, W, C), dtype=TDT[prec], device='cuda')
    lib.call('sfvos_frames_to_ndhwc', P(view), view.stride(1), view.stride(0), view.stride(2), view.stride(3), P(dst),
             dt, T, C, H, W, C, S())
    ref = frames.permute(0, 2, 3, 1)
    assert relmax(dst.float().cpu(), ref) < (1e-7 if prec == 'fp32' else 8e-3)
    back = torch.zeros((T, C, H, W), dtype=torch.float32, device='cuda')
    lib.call('sfvos_ndhwc_to_frames', P(dst), dt, P(back), back.stride(0), back.stride(1), back.stride(2),
             back.stride(3), T, C, H, W, C, 0, S())
    assert torch.equal(back.cpu(), dst.float().cpu().permute(0, 3, 1, 2))
    lib.call('sfvos_ndhwc_to_frames', P(dst), dt, P(back), back.stride(0), back.stride(1), back.stride(2),
             back.stride(3), T, C, H, W, C, 1, S())
    assert torch.equal(back.cpu(), 2 * dst.float().cpu().permute(0, 3, 1, 2))
    M = H * W
    planar = torch.empty((C, M), dtype=torch.float32, device='cuda')
    lib.call('sfvos_ndhwc_to_planar', P(dst[0]), dt, P(planar), M, C, C, S())
    assert torch.equal(planar.cpu(), dst[0].float().cpu().reshape(M, C).t())
    again = torch.empty((M, C), dtype=TDT[prec], device='cuda')
    lib.call('sfvos_planar_to_ndhwc', P(planar), P(again), dt, M, C, C, S())
    assert torch.equal(again.cpu(), dst[0].reshape(M, C).cpu())


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('C,relu', [(192, 1), (224, 0), (32, 1), (64, 1)])
def test_batchnorm_forward_backward(lib, prec, C, relu):
    M = 3 * 11 * 23
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(M, C, generator=g) * 1.7 + 0.3)
    dy = torch.randn(M, C, generator=g)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    if prec == 'bf16':
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    dt = lib.F32 if prec == 'fp32' else lib.BF16
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    yr = F.batch_norm(xr.t().reshape(1, C, M), rm, rv, gr, br, True, 0.1, 1e-5)
    yr = F.relu(yr) if relu else yr
    yr.backward(dy.t().reshape(1, C, M))
    # forward statistics from a single "partial row" (sum, sumsq)
    part = torch.stack([x.double().sum(0), (x.double() ** 2).sum(0)]).float().reshape(1, 2, C).cuda()
    cf = torch.empty((5, C), dtype=torch.float32, device='cuda')
    gd, bd = gamma.cuda(), beta.cuda()
    lib.call('sfvos_bn_finalize', P(part), 1, M, P(gd), P(bd), 1e-5, C, P(cf[0]), P(cf[1]), P(cf[2]), P(cf[3]),
             P(cf[4]), S())
    rmd, rvd = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    lib.call('sfvos_bn_running_update', P(rmd), P(rvd), P(cf[0]), P(cf[4]), 1, C, 0.1, S())
    assert relmax(rmd.cpu(), rm) < 1e-5 and relmax(rvd.cpu(), rv) < 1e-5
    ld = C + 32
    xd = torch.zeros((M, C), dtype=TDT[prec], device='cuda').copy_(x)
    y = torch.full((M, ld), 9.0, dtype=TDT[prec], device='cuda')
    lib.call('sfvos_bn_apply', P(xd), C, P(y, 32), ld, dt, M, C, P(cf[2]), P(cf[3]), relu, S())
    ref_y = yr.detach().reshape(C, M).t()
    assert relmax(y[:, 32:].float().cpu(), ref_y) < (1e-5 if prec == 'fp32' else 8e-3)
    assert torch.all(y[:, :32].float() == 9.0)
    # backward
    dyd = torch.zeros((M, ld), dtype=TDT[prec], device='cuda')
    dyd[:, 32:] = dy.to(TDT[prec])
    rows = lib.load().sfvos_bn_bwd_rows(M)
    bpart = torch.empty((rows, 2, C), dtype=torch.float32, device='cuda')
    lib.call('sfvos_bn_bwd_reduce', P(dyd, 32), ld, P(xd), C, dt, M, C, P(cf[2]), P(cf[3]), P(cf[0]), P(cf[1]), relu,
             P(bpart), S())
    dg, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    abk = torch.empty((3, C), device='cuda')
    lib.call('sfvos_bn_bwd_finalize', P(bpart), rows, M, P(gd), P(cf[0]), P(cf[1]), C, 1, 0, P(dg), P(db), P(abk[0]),
             P(abk[1]), P(abk[2]), S())
    assert relmax(dg.cpu(), gr.grad) < 1e-4 and relmax(db.cpu(), br.grad) < 1e-4
    dx = torch.empty((M, C), dtype=TDT[prec], device='cuda')
    biasp = torch.empty((rows, C), dtype=torch.float32, device='cuda')
    lib.call('sfvos_bn_bwd_apply', P(dyd, 32), ld, P(xd), C, P(dx), C, dt, M, C, P(cf[2]), P(cf[3]), relu, P(abk[0]),
             P(abk[1]), P(abk[2]), P(biasp), S())
    assert relmax(dx.float().cpu(), xr.grad) < (1e-4 if prec == 'fp32' else 1e-2)
    dbias = torch.empty(C, device='cuda')
    lib.call('sfvos_reduce_rows', P(biasp), rows, C, P(dbias), 0, S())
    assert float(dbias.abs().max()) < 1e-2 * float(dx.float().abs().sum(0).max())  # sums to ~0 in train mode
    if prec == 'fp32':  # (bf16: the kernel sums dx before rounding it for storage)
        assert relmax(dbias.cpu() + 1.0, dx.float().sum(0).cpu() + 1.0) < 1e-3


def test_sgd_step_and_scale(lib):
    n = 100003
    g = torch.Generator().manual_seed(5)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=1e-3, momentum=0.9, weight_decay=1e-4)
    pd, gd, buf = p.cuda(), gr.cuda(), torch.zeros(n, device='cuda')
    for step in range(3):
        pr.grad = gr.clone()
        opt.step()
        lib.call('sfvos_sgd_step', P(pd), P(gd), P(buf), n, 1e-3, 0.9, 1e-4, 1 if step == 0 else 0, S())
        assert float((pd.cpu() - pr.detach()).abs().max()) < 1e-6
    lib.call('sfvos_scale', P(gd), n, 0.25, S())
    assert torch.equal(gd.cpu(), gr * 0.25)


def test_bad_arguments_report_errors(lib):
    d, _ = make_desc(lib, 'fp32', 1, 4, 8, 8, 48, 32, 2, 9, 0, 48, 32)  # c_in not a multiple of 32
    assert lib.load().sfvos_conv3d_stat_rows(ctypes.byref(d)) < 0
    assert b'multiples of 32' in lib.load().sfvos_last_error()
    with pytest.raises(RuntimeError):
        lib.call('sfvos_scale', None, 10, 1.0, S())
