"""MultiScaleRoIAlign on libsfvos kernels: the step between the fused feature maps `SlowFastLayers` returns and the mask
branch.

The reference hands `slow_fast_features` to torchvision's `roi_heads` (code/helpers/model.py:346), whose mask path begins
with `mask_roi_pool = MultiScaleRoIAlign(featmap_names=['0','1','2','3'], output_size=14, sampling_ratio=2)`: a level
mapper (k = floor(4 + log2(sqrt(area)/224) + 1e-6), clamped to the levels present) and `torchvision.ops.roi_align`
(aligned=False) per level.  Same constructor and call signature here; forward and backward run in csrc/roialign.hip (the
backward is a deterministic gather instead of torchvision's atomic scatter), so gradients of the mask loss flow back
into the fused maps and from there into SlowFastLayers.

torchvision is third-party, not vendored and not installed in this build's environment: the arithmetic restates its
published sources and is checked against oracle/roi_align_ref.py -- PARITY UNPINNED BY THE REFERENCE.  No CPU fallback."""
import ctypes
import math

import torch
from torch import nn

from . import _lib
from .module import _ptr, _stream


def _check(t, what):
    if not t.is_cuda:
        raise RuntimeError('%s runs on the GPU through libsfvos.so (no CPU fallback)' % what)
    _lib.load()


def _workspace(n, dev):
    return torch.empty(_lib.load().sfvos_roi_align_workspace_bytes(int(n)), dtype=torch.uint8, device=dev)


class _RoIAlignFn(torch.autograd.Function):
    """feats: one [B,C,H,W] fp32 tensor per level; rois [K,5]; levels int32 [K] (None with a single level)."""

    @staticmethod
    def forward(ctx, rois, levels, scales, pooled, sampling_ratio, *feats):
        K = rois.shape[0]
        C = feats[0].shape[1]
        dev = feats[0].device
        out = torch.empty((K, C, pooled, pooled), dtype=torch.float32, device=dev)
        ws = _workspace(K, dev)
        fs = []
        for lvl, (f, sc) in enumerate(zip(feats, scales)):
            fc = f.detach()
            if fc.dtype != torch.float32 or not fc.is_contiguous():
                fc = fc.float().contiguous()
            fs.append(fc)
            B, _, H, W = fc.shape
            _lib.call('sfvos_roi_align', _ptr(fc), B, C, H, W, _ptr(rois), _ptr(levels) if levels is not None else None,
                      lvl, K, float(sc), pooled, sampling_ratio, _ptr(ws), _ptr(out), _stream())
        ctx.rois, ctx.levels, ctx.scales, ctx.pooled, ctx.sr = rois, levels, scales, pooled, sampling_ratio
        ctx.shapes = [tuple(f.shape) for f in fs]
        return out

    @staticmethod
    def backward(ctx, dout):
        K = ctx.rois.shape[0]
        d = dout.detach()
        if d.dtype != torch.float32 or not d.is_contiguous():
            d = d.float().contiguous()
        ws = _workspace(K, d.device)
        grads = []
        for lvl, (shape, sc) in enumerate(zip(ctx.shapes, ctx.scales)):
            if not ctx.needs_input_grad[5 + lvl]:
                grads.append(None)
                continue
            B, C, H, W = shape
            g = torch.empty(shape, dtype=torch.float32, device=d.device)
            _lib.call('sfvos_roi_align_bwd', _ptr(d) if K else None, B, C, H, W, _ptr(ctx.rois) if K else None,
                      _ptr(ctx.levels) if ctx.levels is not None else None, lvl, K, float(sc), ctx.pooled, ctx.sr,
                      _ptr(ws), _ptr(g), 0, _stream())
            grads.append(g)
        return (None, None, None, None, None) + tuple(grads)


def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """torchvision.ops.roi_align for the configuration the reference uses: boxes = Tensor[K,5] (batch index, x1, y1, x2,
    y2) or a list of Tensor[L,4] per image; aligned=False; sampling_ratio > 0."""
    if aligned:
        raise NotImplementedError('roi_align: aligned=True is not implemented (torchvision\'s Mask R-CNN uses False)')
    if sampling_ratio <= 0:
        raise NotImplementedError('roi_align: adaptive sampling (sampling_ratio <= 0) is not implemented')
    _check(input, 'roi_align')
    pooled = output_size if isinstance(output_size, int) else output_size[0]
    if not isinstance(output_size, int) and output_size[0] != output_size[1]:
        raise NotImplementedError('roi_align: square outputs only')
    rois = boxes if torch.is_tensor(boxes) else convert_to_roi_format(boxes)
    rois = rois.detach().to(device=input.device, dtype=torch.float32).contiguous()
    return _RoIAlignFn.apply(rois, None, (float(spatial_scale),), int(pooled), int(sampling_ratio), input)


def convert_to_roi_format(boxes):
    """poolers.MultiScaleRoIAlign.convert_to_roi_format: list of [L_i,4] per image -> [K,5] with the image index first."""
    cat = torch.cat(list(boxes), 0)
    ids = torch.cat([torch.full_like(b[:, :1], i) for i, b in enumerate(boxes)], 0)
    return torch.cat([ids, cat], 1)


class MultiScaleRoIAlign(nn.Module):
    """torchvision.ops.MultiScaleRoIAlign(featmap_names, output_size, sampling_ratio).

    forward(x: OrderedDict[str, Tensor[B,C,H_l,W_l]], boxes: List[Tensor[L_i,4]], image_shapes: List[(H, W)])
    -> Tensor[K, C, output_size, output_size]; differentiable w.r.t. the feature maps."""

    def __init__(self, featmap_names=('0', '1', '2', '3'), output_size=14, sampling_ratio=2,
                 canonical_scale=224, canonical_level=4):
        super(MultiScaleRoIAlign, self).__init__()
        self.featmap_names = list(featmap_names)
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        if self.output_size[0] != self.output_size[1]:
            raise NotImplementedError('MultiScaleRoIAlign: square outputs only')
        if sampling_ratio <= 0:
            raise NotImplementedError('MultiScaleRoIAlign: adaptive sampling (sampling_ratio <= 0) is not implemented')
        self.sampling_ratio = int(sampling_ratio)
        self.canonical_scale, self.canonical_level = canonical_scale, canonical_level
        self.scales = None
        self.k_min = self.k_max = None

    @staticmethod
    def infer_scale(feature, original_size):
        """poolers.infer_scale: the power of two nearest to feature size / image size (both axes must agree)."""
        scales = []
        for s1, s2 in zip(feature.shape[-2:], original_size):
            scales.append(2.0 ** round(math.log2(float(s1) / float(s2))))
        if scales[0] != scales[1]:
            raise RuntimeError('MultiScaleRoIAlign: the two axes of a feature map imply different scales %s' % (scales,))
        return scales[0]

    def setup_scales(self, features, image_shapes):
        original = [max(s[0] for s in image_shapes), max(s[1] for s in image_shapes)]
        self.scales = [self.infer_scale(f, original) for f in features]
        self.k_min, self.k_max = int(-math.log2(self.scales[0])), int(-math.log2(self.scales[-1]))

    def forward(self, x, boxes, image_shapes):
        feats = [v for k, v in x.items() if k in self.featmap_names]
        if not feats:
            raise RuntimeError('MultiScaleRoIAlign: none of the feature maps %s present' % (self.featmap_names,))
        _check(feats[0], 'MultiScaleRoIAlign')
        if self.scales is None or len(self.scales) != len(feats):
            self.setup_scales(feats, image_shapes)
        dev = feats[0].device
        rois = convert_to_roi_format([b.detach().to(device=dev, dtype=torch.float32) for b in boxes]).contiguous()
        K = rois.shape[0]
        levels = None
        if len(feats) > 1:
            levels = torch.zeros(K, dtype=torch.int32, device=dev)
            _lib.call('sfvos_roi_levels', _ptr(rois) if K else None, K, self.k_min, self.k_max,
                      float(self.canonical_scale), float(self.canonical_level), 1e-6, _ptr(levels) if K else None,
                      _stream())
        return _RoIAlignFn.apply(rois, levels, tuple(self.scales), int(self.output_size[0]), self.sampling_ratio, *feats)
