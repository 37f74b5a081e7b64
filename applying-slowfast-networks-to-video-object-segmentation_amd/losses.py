"""Stand-in training loss for the benchmark and the parity tests.

The reference's loss is the sum of torchvision's RoI-head losses on the fused maps (code/helpers/model.py:346-368);
torchvision is absent here and the heads are out of scope, so the bench and the fixtures use
    loss = sum over FPN levels of mean((out_l - target_l) ** 2)
(`oracle/slowfast_ref.py::proxy_loss` is the CPU statement of the same functional).  Value and gradient are one
libsfvos launch each over all levels (sfvos_mse_loss / sfvos_mse_loss_grad); nothing here touches `oracle/`."""
import ctypes

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _table(outs, targets, grads=None):
    tab = _lib.MseTable()
    tab.n = len(outs)
    for i, (o, t) in enumerate(zip(outs, targets)):
        tab.out[i], tab.target[i], tab.numel[i] = o.data_ptr(), t.data_ptr(), o.numel()
        tab.grad[i] = grads[i].data_ptr() if grads is not None else None
    return tab


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, *tensors):
        outs = [t.detach() for t in tensors[:n]]
        targets = list(tensors[n:])
        for o, t in zip(outs, targets):
            if not o.is_cuda:
                raise RuntimeError('MSEProxyLoss runs on the GPU through libsfvos.so (no CPU fallback)')
            if o.dtype != torch.float32 or t.dtype != torch.float32 or o.shape != t.shape:
                raise RuntimeError('MSEProxyLoss: outputs and targets must be fp32 tensors of equal shape')
        outs = [o if o.is_contiguous() else o.contiguous() for o in outs]
        tab = _table(outs, targets)
        rows = _lib.load().sfvos_mse_loss_rows(ctypes.byref(tab))
        dev = outs[0].device
        part = torch.empty(rows, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.call('sfvos_mse_loss', ctypes.byref(tab), _ptr(part), _ptr(loss), st)
        ctx.outs, ctx.targets, ctx.n = outs, targets, n
        return loss

    @staticmethod
    def backward(ctx, g):
        outs, targets = ctx.outs, ctx.targets
        grads = [torch.empty_like(o) for o in outs]
        tab = _table(outs, targets, grads)
        g = g.detach()
        up = g if (g.dtype == torch.float32 and g.is_cuda) else g.to(outs[0].device, torch.float32)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.call('sfvos_mse_loss_grad', ctypes.byref(tab), _ptr(up), st)
        return (None,) + tuple(grads) + (None,) * ctx.n


class MSEProxyLoss(object):
    """loss(merged) = sum_l mean((merged[l] - target[l])**2) for an OrderedDict level -> [B,256,H,W] fp32 (the return
    value of temporally_enhance_features / enhance_packed); targets: dict level -> tensor of the same shape."""

    def __init__(self, targets):
        self.targets = {k: v.contiguous() for k, v in targets.items()}

    def __call__(self, merged):
        keys = list(merged.keys())
        if len(keys) > _lib.MAX_LEVELS:
            raise RuntimeError('at most %d levels' % _lib.MAX_LEVELS)
        return _MseFn.apply(len(keys), *([merged[k] for k in keys] + [self.targets[k] for k in keys]))
