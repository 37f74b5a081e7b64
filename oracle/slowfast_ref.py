"""CPU restatement of the reference's SlowFastLayers (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/code/helpers/model.py:
  * kernel-size tables                      model.py:96-109
  * module order / shapes / state-dict keys model.py:37-69, 71-94
  * fuse (lateral conv -> BN -> ReLU -> cat) model.py:111-116
  * forward graph                           model.py:118-149
  * temporally_enhance_features             model.py:151-165
  * SGD(momentum, weight_decay) as train.py:80 builds it, stepped every 2nd
    clip as model.py:369-374 does.

The arithmetic is torch-CPU fp32 (F.conv3d / F.batch_norm / relu / cat), i.e.
the same third-party ATen kernels the reference dispatches to; parity with the
reference's own class is pinned by tests/golden (see oracle/make_golden.py).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


def temporal_kernel_sizes(pathway_size):
    """model.py:96-103 -- three valid temporal kernels that take T from p to 1."""
    d, r = divmod(pathway_size, 3)
    if r == 0:
        return (d, d + 1, d + 1)
    if r == 1:
        return (d + 1, d + 1, d + 1)
    return (d + 1, d + 1, d + 2)


def lateral_kernel_size(slow_in, slow_k, fast_in, fast_k):
    """model.py:105-109 -- k such that a valid conv maps fast's T' onto slow's T'."""
    out_slow = slow_in - slow_k + 1
    out_fast = fast_in - fast_k + 1
    return out_fast - out_slow + 1, out_slow, out_fast


# (name, bn name, cin, cout) in the reference's registration order (model.py:47-67)
_MAIN = [('fast_conv1', 'bn_f1', None, 32), ('slow_conv1', 'bn_s1', None, 192),
         ('fast_conv2', 'bn_f2', 32, 32), ('slow_conv2', 'bn_s2', 256, 192),
         ('fast_conv3', 'bn_f3', 32, 32), ('slow_conv3', 'bn_s3', 256, 224)]


class OracleSlowFastLayers(nn.Module):
    """Same constructor, methods and state-dict keys as the reference class."""

    def __init__(self, input_size, device, slow_pathway_size, fast_pathway_size):
        super().__init__()
        self.device = device
        self.slow_pathway_size = slow_pathway_size
        self.fast_pathway_size = fast_pathway_size
        ks = temporal_kernel_sizes(slow_pathway_size)
        kf = temporal_kernel_sizes(fast_pathway_size)
        l1, so1, fo1 = lateral_kernel_size(slow_pathway_size, ks[0], fast_pathway_size, kf[0])
        l2, _, _ = lateral_kernel_size(so1, ks[1], fo1, kf[1])
        self.kernel_sizes = {'slow': ks, 'fast': kf, 'lateral': (l1, l2)}
        # tests may set this to a list: every ReLU then appends min |input| (how far the clip is from a mask flip
        # between two correct implementations -- the ReLU derivative is discontinuous at 0)
        self.relu_margins = None
        # tests may set relu_masks = {(level key, bn name): bool [B,C,T,H,W]}: the ReLU of that layer then uses the GIVEN
        # mask (y * mask) instead of its own (y > 0), and relu_flips[(level key, bn name)] counts the elements whose
        # own mask differs -- the experiment behind DESIGN.md's "ReLU-mask flips" (two correct implementations put a
        # pre-activation within round-off of 0 on different sides).  Default None: the reference's arithmetic, untouched.
        self.relu_masks = None
        self.relu_flips = {}
        self._level_key = None
        kt = {'fast_conv1': kf[0], 'slow_conv1': ks[0], 'fast_conv2': kf[1], 'slow_conv2': ks[1],
              'fast_conv3': kf[2], 'slow_conv3': ks[2]}
        for conv, bn, cin, cout in _MAIN:
            cin = input_size if cin is None else cin
            self.add_module(conv, nn.Conv3d(cin, cout, (kt[conv], 3, 3), padding=(0, 1, 1)))
            self.add_module(bn, nn.BatchNorm3d(cout))
        for i, k in ((1, l1), (2, l2)):
            self.add_module('conv_f2s%d' % i, nn.Conv3d(32, 64, (k, 1, 1), bias=False))
            self.add_module('bn_f2s%d' % i, nn.BatchNorm3d(64))

    # -- functional pieces -------------------------------------------------
    def _conv_bn(self, x, conv, bn, relu, pad):
        c = getattr(self, conv)
        b = getattr(self, bn)
        y = F.conv3d(x, c.weight, c.bias, stride=1, padding=pad)
        if self.training:
            b.num_batches_tracked += 1
        y = F.batch_norm(y, b.running_mean, b.running_var, b.weight, b.bias,
                         training=self.training, momentum=0.1, eps=1e-5)
        if relu and self.relu_margins is not None:
            self.relu_margins.append(float(y.detach().abs().min()))
        if relu and self.relu_masks is not None and (self._level_key, bn) in self.relu_masks:
            mask = self.relu_masks[(self._level_key, bn)]
            self.relu_flips[(self._level_key, bn)] = int(((y.detach() > 0) != mask).sum())
            return y * mask.to(y.dtype)
        return F.relu(y) if relu else y

    def forward(self, slow, fast):
        s = self._conv_bn(slow, 'slow_conv1', 'bn_s1', True, (0, 1, 1))
        f = self._conv_bn(fast, 'fast_conv1', 'bn_f1', True, (0, 1, 1))
        s = torch.cat([s, self._conv_bn(f, 'conv_f2s1', 'bn_f2s1', True, 0)], 1)
        s = self._conv_bn(s, 'slow_conv2', 'bn_s2', True, (0, 1, 1))
        f = self._conv_bn(f, 'fast_conv2', 'bn_f2', True, (0, 1, 1))
        s = torch.cat([s, self._conv_bn(f, 'conv_f2s2', 'bn_f2s2', True, 0)], 1)
        s = self._conv_bn(s, 'slow_conv3', 'bn_s3', False, (0, 1, 1))
        f = self._conv_bn(f, 'fast_conv3', 'bn_f3', False, (0, 1, 1))
        return s, f

    def temporally_enhance_features(self, slow_features, fast_features):
        merged = OrderedDict()
        for key in slow_features[0].keys():
            self._level_key = key
            s = torch.stack([d[key] for d in slow_features]).to(self.device).transpose(1, 2)
            f = torch.stack([d[key] for d in fast_features]).to(self.device).transpose(1, 2)
            s, f = self.forward(s, f)
            merged[key] = torch.cat([s, f], dim=1).squeeze(dim=2)
        return merged


def proxy_loss(merged):
    """Stand-in for the RoI-head losses (torchvision, absent): sum over levels of
    mean((out - target)**2) with a fixed closed-form target per level.

    SURVEY.md 8c/8d proposed sum_l mean(out_l**2); with train-mode BatchNorm as the last op of both
    pathways that functional is (numerically) constant in everything upstream -- its conv-weight
    gradients are pure round-off -- so it cannot pin a backward pass.  The target makes the gradient
    of every parameter O(1e-3) and well conditioned.  Defined by this build, not by the reference."""
    total = None
    for k, v in merged.items():
        tgt = _target(k, tuple(v.shape), v.device)
        term = ((v.float() - tgt) ** 2).mean()
        total = term if total is None else total + term
    return total


_TARGETS = {}


def _target(key, shape, device):
    """closed-form target of one level, built once per (level, shape, device)."""
    from oracle.closed_form import closed_form_tensor
    ck = (key, shape, str(device))
    if ck not in _TARGETS:
        _TARGETS[ck] = closed_form_tensor(shape, 'target/%s' % key, 4.0).to(device)
    return _TARGETS[ck]


def proxy_argmax(merged):
    """Per-pixel argmax over the 256 fused channels: the build-defined discrete
    proxy for 'mask indices' (the reference has no argmax)."""
    return OrderedDict((k, v.argmax(dim=1)) for k, v in merged.items())


def sgd_step_(params, momentum_bufs, lr=1e-3, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD semantics as configured by reference train.py:80
    (dampening 0, no nesterov): g += wd*p ; buf = g (first step) or m*buf + g ; p -= lr*buf."""
    with torch.no_grad():
        for i, p in enumerate(params):
            if p.grad is None:
                continue
            g = p.grad + weight_decay * p
            if momentum_bufs[i] is None:
                momentum_bufs[i] = g.clone()
            else:
                momentum_bufs[i].mul_(momentum).add_(g)
            p.add_(momentum_bufs[i], alpha=-lr)
