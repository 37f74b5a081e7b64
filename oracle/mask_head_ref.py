"""CPU restatement of the Mask R-CNN mask branch behind RoIAlign (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the arithmetic lives in torchvision (`code/requirements.txt:2` torchvision>=0.6.0), which is not
vendored in /root/reference and not installed here, and no reference test or fixture covers it.  This file restates
the published torchvision modules the reference instantiates / calls (code/helpers/model.py:17-25, 346-347) with
torch-core ops:
  * models/detection/mask_rcnn.py  MaskRCNNHeads     4 x [Conv2d(256,256,3,1,1) + ReLU], kaiming_normal_(fan_out, relu)
  * models/detection/mask_rcnn.py  MaskRCNNPredictor  ConvTranspose2d(256,256,2,2,0) + ReLU + Conv2d(256,K,1,1,0)
  * models/detection/roi_heads.py  maskrcnn_inference sigmoid, pick the label's channel
  * models/detection/roi_heads.py  maskrcnn_loss      BCE with logits on the label's channel (training, model.py:369)
  * models/detection/roi_heads.py  expand_masks / expand_boxes / paste_mask_in_image / paste_masks_in_image
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


class OracleMaskRCNNHeads(nn.Sequential):
    def __init__(self, in_channels=256, layers=(256, 256, 256, 256), dilation=1):
        d = OrderedDict()
        nf = in_channels
        for i, feat in enumerate(layers, 1):
            d['mask_fcn%d' % i] = nn.Conv2d(nf, feat, kernel_size=3, stride=1, padding=dilation, dilation=dilation)
            d['relu%d' % i] = nn.ReLU(inplace=True)
            nf = feat
        super().__init__(d)
        for name, p in self.named_parameters():
            if 'weight' in name:
                nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')


class OracleMaskRCNNPredictor(nn.Sequential):
    def __init__(self, in_channels=256, dim_reduced=256, num_classes=2):
        super().__init__(OrderedDict([
            ('conv5_mask', nn.ConvTranspose2d(in_channels, dim_reduced, 2, 2, 0)),
            ('relu', nn.ReLU(inplace=True)),
            ('mask_fcn_logits', nn.Conv2d(dim_reduced, num_classes, 1, 1, 0))]))
        for name, p in self.named_parameters():
            if 'weight' in name:
                nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')


class OracleMaskBranch(nn.Module):
    def __init__(self, in_channels=256, num_classes=2):
        super().__init__()
        self.mask_head = OracleMaskRCNNHeads(in_channels)
        self.mask_predictor = OracleMaskRCNNPredictor(256, 256, num_classes)

    def forward(self, x):
        return self.mask_predictor(self.mask_head(x))


def maskrcnn_loss(mask_logits, labels, mask_targets):
    """roi_heads.maskrcnn_loss behind project_masks_on_boxes (the RoIAlign of the ground-truth masks, torchvision's):
    BCE with logits between the label's channel and the [N,M,M] targets; `sum * 0` without positive RoIs."""
    if mask_targets.numel() == 0:
        return mask_logits.sum() * 0
    index = torch.arange(labels.shape[0], device=labels.device)
    return F.binary_cross_entropy_with_logits(mask_logits[index, labels], mask_targets)


def maskrcnn_inference(x, labels):
    """roi_heads.maskrcnn_inference for one image: [N,K,M,M] logits -> [N,1,M,M] probabilities of the label's class."""
    mask_prob = x.sigmoid()
    index = torch.arange(mask_prob.shape[0], device=labels.device)
    return mask_prob[index, labels][:, None]


def expand_boxes(boxes, scale):
    w_half = (boxes[:, 2] - boxes[:, 0]) * .5
    h_half = (boxes[:, 3] - boxes[:, 1]) * .5
    x_c = (boxes[:, 2] + boxes[:, 0]) * .5
    y_c = (boxes[:, 3] + boxes[:, 1]) * .5
    w_half = w_half * scale
    h_half = h_half * scale
    out = torch.zeros_like(boxes)
    out[:, 0] = x_c - w_half
    out[:, 2] = x_c + w_half
    out[:, 1] = y_c - h_half
    out[:, 3] = y_c + h_half
    return out


def expand_masks(mask, padding):
    M = mask.shape[-1]
    scale = float(M + 2 * padding) / M
    return F.pad(mask, (padding,) * 4), scale


def paste_mask_in_image(mask, box, im_h, im_w):
    TO_REMOVE = 1
    w = max(int(box[2] - box[0] + TO_REMOVE), 1)
    h = max(int(box[3] - box[1] + TO_REMOVE), 1)
    mask = mask.expand((1, 1, -1, -1))
    mask = F.interpolate(mask, size=(h, w), mode='bilinear', align_corners=False)[0][0]
    im_mask = torch.zeros((im_h, im_w), dtype=mask.dtype, device=mask.device)
    x_0 = max(int(box[0]), 0)
    x_1 = min(int(box[2]) + 1, im_w)
    y_0 = max(int(box[1]), 0)
    y_1 = min(int(box[3]) + 1, im_h)
    if x_1 > x_0 and y_1 > y_0:
        im_mask[y_0:y_1, x_0:x_1] = mask[(y_0 - int(box[1])):(y_1 - int(box[1])), (x_0 - int(box[0])):(x_1 - int(box[0]))]
    return im_mask


def paste_masks_in_image(masks, boxes, img_shape, padding=1):
    masks, scale = expand_masks(masks, padding=padding)
    boxes = expand_boxes(boxes, scale).to(dtype=torch.int64)
    im_h, im_w = img_shape
    res = [paste_mask_in_image(m[0], b, im_h, im_w) for m, b in zip(masks, boxes)]
    if len(res) > 0:
        return torch.stack(res, dim=0)[:, None]
    return masks.new_empty((0, 1, im_h, im_w))
