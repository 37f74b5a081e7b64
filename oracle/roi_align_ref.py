"""CPU restatement of torchvision's MultiScaleRoIAlign (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the arithmetic lives in torchvision (`code/requirements.txt:2` torchvision>=0.6.0), which is neither
vendored in /root/reference nor installed here, and no reference test or fixture covers it.  The reference reaches it
through `self.maskrcnn_model.roi_heads(slow_fast_features, proposals, image_sizes, targets)` (code/helpers/model.py:346):
roi_heads' mask path is `mask_roi_pool = MultiScaleRoIAlign(featmap_names=['0','1','2','3'], output_size=14,
sampling_ratio=2)` (torchvision.models.detection.mask_rcnn.MaskRCNN defaults), then mask_head, mask_predictor.

Restated from the published sources with torch-core ops (fp32 arithmetic in torchvision's order):
  * ops/poolers.py   LevelMapper (canonical_scale 224, canonical_level 4, eps 1e-6), infer_scale / setup_scales,
                     convert_to_roi_format, MultiScaleRoIAlign.forward
  * ops/csrc/cpu/roi_align_kernel.cpp  roi_align forward (aligned = False): bilinear_interpolate's rules -- a sample more
                     than one pixel outside the map contributes 0; coordinates clamp at 0; the last row / column
                     interpolates with itself -- and the mean over sampling_ratio^2 samples per bin.
The backward used as reference is torch autograd through this restatement (torchvision's own backward scatters the same
weights with atomics)."""
import math

import torch


def _axis_samples(start, extent, pooled, grid, size):
    """Sample coordinates of one axis -> (lo, hi, wlo, whi, valid): fp32, torchvision's operation order."""
    bin_size = extent / pooled
    p = torch.arange(pooled, dtype=torch.float32).repeat_interleave(grid)
    i = torch.arange(grid, dtype=torch.float32).repeat(pooled)
    y = start + p * bin_size + (i + 0.5) * bin_size / float(grid)
    valid = ~((y < -1.0) | (y > float(size)))
    y = torch.where(y <= 0, torch.zeros_like(y), y)
    lo = y.to(torch.int64)
    edge = lo >= size - 1
    lo = torch.where(edge, torch.full_like(lo, size - 1), lo)
    hi = torch.where(edge, lo, lo + 1)
    y = torch.where(edge, lo.to(torch.float32), y)
    l = y - lo.to(torch.float32)
    return lo.clamp(0, size - 1), hi.clamp(0, size - 1), 1.0 - l, l, valid


def roi_align(feat, rois, output_size, spatial_scale, sampling_ratio):
    """torchvision.ops.roi_align(feat [B,C,H,W], rois [K,5], output_size, spatial_scale, sampling_ratio, aligned=False)
    -> [K,C,P,P]; differentiable w.r.t. feat."""
    P = int(output_size)
    g = int(sampling_ratio)
    assert g > 0, 'adaptive sampling (sampling_ratio <= 0) is not restated: the reference configuration uses 2'
    B, C, H, W = feat.shape
    out = []
    scale = torch.tensor(spatial_scale, dtype=torch.float32)
    one = torch.tensor(1.0)
    for roi in rois.detach().to(torch.float32):
        b = int(roi[0])
        sw, sh, ew, eh = roi[1] * scale, roi[2] * scale, roi[3] * scale, roi[4] * scale
        rw, rh = torch.maximum(ew - sw, one), torch.maximum(eh - sh, one)
        ylo, yhi, wyl, wyh, vy = _axis_samples(sh, rh, P, g, H)
        xlo, xhi, wxl, wxh, vx = _axis_samples(sw, rw, P, g, W)
        f = feat[b]
        v1 = f[:, ylo[:, None], xlo[None, :]]
        v2 = f[:, ylo[:, None], xhi[None, :]]
        v3 = f[:, yhi[:, None], xlo[None, :]]
        v4 = f[:, yhi[:, None], xhi[None, :]]
        w1, w2 = wyl[:, None] * wxl[None, :], wyl[:, None] * wxh[None, :]
        w3, w4 = wyh[:, None] * wxl[None, :], wyh[:, None] * wxh[None, :]
        val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
        val = val * (vy[:, None] & vx[None, :]).to(val.dtype)
        out.append(val.reshape(C, P, g, P, g).sum((2, 4)) / float(max(g * g, 1)))
    if not out:
        return feat.new_zeros((0, C, P, P))
    return torch.stack(out)


class LevelMapper(object):
    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min, self.k_max, self.s0, self.lvl0, self.eps = k_min, k_max, canonical_scale, canonical_level, eps

    def __call__(self, boxlists):
        s = torch.sqrt(torch.cat([(b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) for b in boxlists]))
        target = torch.floor(self.lvl0 + torch.log2(s / self.s0) + torch.tensor(self.eps, dtype=s.dtype))
        target = torch.clamp(target, min=self.k_min, max=self.k_max)
        return (target.to(torch.int64) - self.k_min).to(torch.int64)


class OracleMultiScaleRoIAlign(object):
    def __init__(self, featmap_names=('0', '1', '2', '3'), output_size=14, sampling_ratio=2):
        self.featmap_names, self.output_size, self.sampling_ratio = list(featmap_names), output_size, sampling_ratio

    @staticmethod
    def infer_scale(feature, original_size):
        scales = []
        for s1, s2 in zip(feature.shape[-2:], original_size):
            scales.append(2 ** float(torch.tensor(float(s1) / float(s2)).log2().round()))
        assert scales[0] == scales[1]
        return scales[0]

    def __call__(self, x, boxes, image_shapes):
        feats = [v for k, v in x.items() if k in self.featmap_names]
        original = [max(s[0] for s in image_shapes), max(s[1] for s in image_shapes)]
        scales = [self.infer_scale(f, original) for f in feats]
        mapper = LevelMapper(int(-math.log2(scales[0])), int(-math.log2(scales[-1])))
        ids = torch.cat([torch.full((len(b), 1), i, dtype=torch.float32) for i, b in enumerate(boxes)])
        rois = torch.cat([ids, torch.cat(boxes).to(torch.float32)], 1)
        if len(feats) == 1:
            return roi_align(feats[0], rois, self.output_size, scales[0], self.sampling_ratio)
        levels = mapper(boxes)
        C = feats[0].shape[1]
        result = torch.zeros((rois.shape[0], C, self.output_size, self.output_size), dtype=feats[0].dtype)
        for lvl, (f, sc) in enumerate(zip(feats, scales)):
            idx = torch.where(levels == lvl)[0]
            if len(idx):
                result = result.index_put((idx,), roi_align(f, rois[idx], self.output_size, sc, self.sampling_ratio))
        return result
