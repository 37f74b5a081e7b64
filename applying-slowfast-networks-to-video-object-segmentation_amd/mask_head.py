"""The post-RoIAlign part of the Mask R-CNN mask branch, on libsfvos kernels.

The reference builds `torchvision.models.detection.maskrcnn_resnet50_fpn` and swaps in
`MaskRCNNPredictor(in_features_mask, 256, num_classes=2)` (code/helpers/model.py:17-25); `roi_heads`
(model.py:346) runs RoIAlign -> `mask_head` (MaskRCNNHeads: 4 x conv3x3 256->256 + ReLU) -> `mask_predictor`
(ConvTranspose2d 2x2 s2 -> ReLU -> conv1x1 -> num_classes) -> `maskrcnn_inference` (sigmoid, the label's channel), and
`transform.postprocess` (model.py:347) pastes the 28x28 masks into the image (`paste_masks_in_image`).

torchvision is third-party, not vendored and absent in this build's environment, so these classes mirror its
published module structure (same attribute names and state-dict keys, so that the slices
`roi_heads.mask_head.*` / `roi_heads.mask_predictor.*` of a reference checkpoint load with strict=True) and their
results are checked against a torch-core restatement (oracle/mask_head_ref.py) -- PARITY UNPINNED BY THE REFERENCE.
RoIAlign itself stays torchvision's (out of scope).  No CPU fallback.

Training: the reference trains roi_heads (only backbone and RPN are frozen, model.py:176-179; losses.backward() at
model.py:369).  `MaskBranch.forward` is differentiable (w.r.t. the RoI features and all 12 parameters) through one
autograd Function whose backward runs on libsfvos kernels, and `maskrcnn_loss` is torchvision's mask loss behind the
RoIAlign of the ground-truth masks (the caller passes the [N,28,28] targets)."""
import ctypes

import torch
from torch import nn

from . import _lib
from .module import _DT, _ptr, _stream


def _check_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError('%s runs on the GPU through libsfvos.so (no CPU fallback)' % what)
    _lib.load()


def _to_nhwc(x, dt_id, tdt):
    """[N,C,H,W] fp32 (any strides) -> NHWC [N,H,W,C] in the compute dtype (sfvos_frames_to_ndhwc with T = N)."""
    N, C, H, W = x.shape
    s = x if x.dtype == torch.float32 else x.float()
    out = torch.empty((N, H, W, C), dtype=tdt, device=x.device)
    _lib.call('sfvos_frames_to_ndhwc', _ptr(s), s.stride(0), s.stride(1), s.stride(2), s.stride(3), _ptr(out), dt_id,
              N, C, H, W, C, _stream())
    return out


def _to_nchw(x_nhwc, dt_id):
    N, H, W, C = x_nhwc.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x_nhwc.device)
    _lib.call('sfvos_ndhwc_to_frames', _ptr(x_nhwc), dt_id, _ptr(out), out.stride(0), out.stride(1), out.stride(2),
              out.stride(3), N, C, H, W, C, 0, _stream())
    return out


class MaskRCNNHeads(nn.Module):
    """torchvision's MaskRCNNHeads(in_channels, layers, dilation=1): mask_fcn{i} = Conv2d(3x3, pad 1) + ReLU.
    forward(x [N,C,H,W] fp32) -> [N,layers[-1],H,W] fp32; `forward_nhwc` keeps the channels-last compute tensor."""

    def __init__(self, in_channels=256, layers=(256, 256, 256, 256), dilation=1, precision='fp32'):
        super(MaskRCNNHeads, self).__init__()
        if dilation != 1:
            raise ValueError('MaskRCNNHeads: only dilation 1 (the reference configuration) is implemented')
        if precision not in ('fp32', 'bf16'):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.precision = precision
        self.n_layers = len(layers)
        nf = in_channels
        for i, feat in enumerate(layers, 1):
            self.add_module('mask_fcn%d' % i, nn.Conv2d(nf, feat, kernel_size=3, stride=1, padding=1))
            self.add_module('relu%d' % i, nn.ReLU(inplace=True))
            nf = feat
        for name, p in self.named_parameters():   # torchvision's init
            if 'weight' in name:
                nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')
        self._packs = {}

    def _packed(self, conv, dt_id, tdt, dgrad=False):
        w = conv.weight
        key = (id(conv), dgrad)
        # weight_epoch: FusedSGD rewrites parameters through raw pointers (no autograd version bump), see module.py
        tag = (w._version, w.data_ptr(), dt_id, _lib.weight_epoch())
        hit = self._packs.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1]
        wc = w.detach().float().contiguous()
        packed = torch.empty(wc.numel(), dtype=tdt, device=w.device)
        _lib.call('sfvos_pack_weights_dgrad' if dgrad else 'sfvos_pack_weights_fwd', _ptr(wc), _ptr(packed), dt_id,
                  conv.out_channels, conv.in_channels, 1, 9, _stream())
        self._packs[key] = (tag, packed)
        return packed

    def forward_nhwc(self, x, keep=False):
        """keep: return every activation [input, after mask_fcn1 + ReLU, ...] (channels-last) for the backward."""
        _check_gpu(x, 'MaskRCNNHeads')
        dt_id, tdt = _DT[self.precision]
        N, C, H, W = x.shape
        cur = _to_nhwc(x.detach(), dt_id, tdt)
        acts = [cur]
        for i in range(1, self.n_layers + 1):
            conv = getattr(self, 'mask_fcn%d' % i)
            d = _conv_desc(dt_id, N, H, W, conv.in_channels, conv.out_channels, 1)
            y = torch.empty((N, H, W, conv.out_channels), dtype=tdt, device=x.device)
            bias = _ptr(conv.bias.detach()) if conv.bias is not None else None
            _lib.call('sfvos_conv3d', ctypes.byref(d), _ptr(cur), _ptr(self._packed(conv, dt_id, tdt)), bias, _ptr(y),
                      None, _stream())
            cur = y
            acts.append(cur)
        return acts if keep else cur

    def forward(self, x):
        dt_id, _ = _DT[self.precision]
        return _to_nchw(self.forward_nhwc(x), dt_id)


class MaskRCNNPredictor(nn.Module):
    """torchvision's MaskRCNNPredictor(in_channels, dim_reduced, num_classes): conv5_mask = ConvTranspose2d(2, 2, 0),
    relu, mask_fcn_logits = Conv2d(1x1).  forward(x [N,C,H,W]) -> logits [N,num_classes,2H,2W] fp32."""

    def __init__(self, in_channels=256, dim_reduced=256, num_classes=2, precision='fp32'):
        super(MaskRCNNPredictor, self).__init__()
        if precision not in ('fp32', 'bf16'):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.precision = precision
        self.conv5_mask = nn.ConvTranspose2d(in_channels, dim_reduced, 2, 2, 0)
        self.relu = nn.ReLU(inplace=True)
        self.mask_fcn_logits = nn.Conv2d(dim_reduced, num_classes, 1, 1, 0)
        for name, p in self.named_parameters():
            if 'weight' in name:
                nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')
        self._pack = None
        self._pack_d = None

    def _deconv_nhwc(self, x_nhwc):
        dt_id, tdt = _DT[self.precision]
        N, H, W, C = x_nhwc.shape
        w = self.conv5_mask.weight
        tag = (w._version, w.data_ptr(), dt_id, _lib.weight_epoch())
        if self._pack is None or self._pack[0] != tag:
            wc = w.detach().float().contiguous()
            packed = torch.empty(wc.numel(), dtype=tdt, device=w.device)
            _lib.call('sfvos_pack_deconv2x2', _ptr(wc), _ptr(packed), dt_id, w.shape[0], w.shape[1], _stream())
            self._pack = (tag, packed)
        y = torch.empty((N, 2 * H, 2 * W, w.shape[1]), dtype=tdt, device=x_nhwc.device)
        b = self.conv5_mask.bias
        _lib.call('sfvos_deconv2x2_relu', _ptr(x_nhwc), _ptr(self._pack[1]), _ptr(b.detach()) if b is not None else None,
                  _ptr(y), dt_id, N, H, W, w.shape[0], w.shape[1], 1, _stream())
        return y

    def _packed_dgrad(self, dt_id, tdt):
        w = self.conv5_mask.weight
        tag = (w._version, w.data_ptr(), dt_id, _lib.weight_epoch())
        if self._pack_d is None or self._pack_d[0] != tag:
            wc = w.detach().float().contiguous()
            packed = torch.empty(wc.numel(), dtype=tdt, device=w.device)
            _lib.call('sfvos_pack_deconv2x2_dgrad', _ptr(wc), _ptr(packed), dt_id, w.shape[0], w.shape[1], _stream())
            self._pack_d = (tag, packed)
        return self._pack_d[1]

    def _logits(self, y_nhwc, labels=None, want_logits=True, want_prob=False):
        dt_id, _ = _DT[self.precision]
        N, H, W, C = y_nhwc.shape
        conv = self.mask_fcn_logits
        K = conv.out_channels
        wl = conv.weight.detach().float().reshape(K, C).contiguous()
        logits = torch.empty((N, K, H, W), dtype=torch.float32, device=y_nhwc.device) if want_logits else None
        prob = torch.empty((N, 1, H, W), dtype=torch.float32, device=y_nhwc.device) if want_prob else None
        lab = None
        if labels is not None:
            lab = labels.to(device=y_nhwc.device, dtype=torch.int64).contiguous()
            if lab.numel() != N:
                raise RuntimeError('one label per RoI expected')
        _lib.call('sfvos_mask_logits', _ptr(y_nhwc), dt_id, _ptr(wl), _ptr(conv.bias.detach()) if conv.bias is not None
                  else None, _ptr(lab) if lab is not None else None, N, H * W, C, K,
                  _ptr(logits) if logits is not None else None, _ptr(prob) if prob is not None else None, _stream())
        return logits, prob

    def forward_from_nhwc(self, x_nhwc, labels=None, want_logits=True, want_prob=False):
        return self._logits(self._deconv_nhwc(x_nhwc), labels, want_logits, want_prob)

    def forward(self, x):
        _check_gpu(x, 'MaskRCNNPredictor')
        dt_id, tdt = _DT[self.precision]
        return self.forward_from_nhwc(_to_nhwc(x.detach(), dt_id, tdt))[0]


def maskrcnn_inference(mask_logits, labels):
    """torchvision roi_heads.maskrcnn_inference for ONE image: sigmoid, then the channel of each box's label ->
    [N,1,M,M].  (When the heads above produced the logits, MaskBranch.predict fuses this into the logits kernel.)"""
    _check_gpu(mask_logits, 'maskrcnn_inference')
    N, K, H, W = mask_logits.shape
    x = mask_logits.detach().float().permute(0, 2, 3, 1).contiguous()        # NHWC view of the K logits
    eye = torch.eye(K, dtype=torch.float32, device=x.device)
    prob = torch.empty((N, 1, H, W), dtype=torch.float32, device=x.device)
    lab = labels.to(device=x.device, dtype=torch.int64).contiguous()
    _lib.call('sfvos_mask_logits', _ptr(x), _lib.F32, _ptr(eye), None, _ptr(lab), N, H * W, K, K, None, _ptr(prob),
              _stream())
    return prob


def paste_masks_in_image(masks, boxes, img_shape, padding=1):
    """torchvision roi_heads.paste_masks_in_image: masks [N,1,M,M] fp32 probabilities, boxes [N,4] (x1,y1,x2,y2) ->
    [N,1,H,W] fp32."""
    im_h, im_w = int(img_shape[0]), int(img_shape[1])
    if masks.shape[0] == 0:
        return masks.new_empty((0, 1, im_h, im_w))
    _check_gpu(masks, 'paste_masks_in_image')
    m = masks.detach().float().contiguous()
    b = boxes.detach().to(device=m.device, dtype=torch.float32).contiguous()
    out = torch.empty((m.shape[0], 1, im_h, im_w), dtype=torch.float32, device=m.device)
    _lib.call('sfvos_paste_masks', _ptr(m), _ptr(b), m.shape[0], m.shape[-1], int(padding), im_h, im_w, _ptr(out),
              _stream())
    return out


def _conv_desc(dt_id, N, H, W, c_in, c_out, relu):
    d = _lib.ConvDesc()
    d.dtype, d.batch, d.t_in, d.t_alloc, d.t_offset = dt_id, N, 1, 1, 0
    d.c_in, d.c_out, d.kt, d.taps, d.pad_t = c_in, c_out, 1, 9, 0
    d.ld_x, d.ld_y, d.accumulate, d.relu, d.pyr = c_in, c_out, 0, relu, _lib.make_pyramid([(H, W)])
    return d


class _MaskBranchFn(torch.autograd.Function):
    """RoI features [N,256,H,W] -> mask logits [N,K,2H,2W], keeping the channels-last activations for the backward.
    Parameter order: mask_fcn{1..L}.(weight, bias), conv5_mask.(weight, bias), mask_fcn_logits.(weight, bias)."""

    @staticmethod
    def forward(ctx, branch, x, *params):
        head, pred = branch.mask_head, branch.mask_predictor
        acts = head.forward_nhwc(x, keep=True)                 # [a0 (input), a1, ..., aL] NHWC, post-ReLU
        y5 = pred._deconv_nhwc(acts[-1])
        logits, _ = pred._logits(y5)
        ctx.branch, ctx.acts, ctx.y5 = branch, acts, y5
        ctx.shape = tuple(x.shape)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        branch, acts, y5 = ctx.branch, ctx.acts, ctx.y5
        head, pred = branch.mask_head, branch.mask_predictor
        dt_id, tdt = _DT[head.precision]
        dev = y5.device
        N, _, H, W = ctx.shape
        f32 = dict(dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad     # (branch, x, *params)
        grads = []
        # ---- mask_fcn_logits + the ReLU in front of it
        conv = pred.mask_fcn_logits
        K, C = conv.out_channels, conv.in_channels
        P2 = 4 * H * W
        dl = dlogits.detach().float().contiguous()
        rows = _lib.load().sfvos_mask_logits_bwd_rows(N, P2)
        L = K * C + K + C
        part = torch.empty((rows, L), **f32)
        dz = torch.empty_like(y5)
        wl = conv.weight.detach().float().reshape(K, C).contiguous()
        _lib.call('sfvos_mask_logits_bwd', _ptr(y5), dt_id, _ptr(dl), _ptr(wl), N, P2, C, K, 1, _ptr(dz), _ptr(part),
                  _stream())
        comb = torch.empty(L, **f32)
        _lib.call('sfvos_reduce_rows', _ptr(part), rows, L, _ptr(comb), 0, _stream())
        g_wl, g_bl, g_b5 = comb[:K * C].view(K, C, 1, 1), comb[K * C:K * C + K], comb[K * C + K:]
        # ---- conv5_mask (ConvTranspose2d 2x2 s2)
        w5 = pred.conv5_mask.weight
        ci5, co5 = w5.shape[0], w5.shape[1]
        g_w5 = torch.empty(w5.shape, **f32)
        ws = torch.empty(_lib.load().sfvos_deconv2x2_wgrad_workspace_bytes(N, H, W, ci5, co5), dtype=torch.uint8,
                         device=dev)
        _lib.call('sfvos_deconv2x2_wgrad', _ptr(acts[-1]), _ptr(dz), dt_id, N, H, W, ci5, co5, _ptr(g_w5), 0, _ptr(ws),
                  _stream())
        cur = torch.empty_like(acts[-1])
        _lib.call('sfvos_deconv2x2_dgrad', _ptr(dz), _ptr(pred._packed_dgrad(dt_id, tdt)), _ptr(cur), dt_id, N, H, W,
                  ci5, co5, _stream())
        tail = [g_w5, g_b5, g_wl, g_bl]
        # ---- mask_fcn{L..1}: ReLU backward (+ bias gradient), weight gradient, data gradient
        M = N * H * W
        lib = _lib.load()
        conv_grads = []
        for i in range(head.n_layers, 0, -1):
            conv = getattr(head, 'mask_fcn%d' % i)
            cin, cout = conv.in_channels, conv.out_channels
            rrows = lib.sfvos_relu_bwd_rows(M)
            rpart = torch.empty((rrows, cout), **f32)
            _lib.call('sfvos_relu_bwd', _ptr(cur), _ptr(acts[i]), _ptr(cur), dt_id, M, cout, _ptr(rpart), _stream())
            g_b = torch.empty(cout, **f32)
            _lib.call('sfvos_reduce_rows', _ptr(rpart), rrows, cout, _ptr(g_b), 0, _stream())
            d = _conv_desc(dt_id, N, H, W, cin, cout, 0)
            g_w = torch.empty(conv.weight.shape, **f32)
            wsz = lib.sfvos_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
            wws = torch.empty(max(int(wsz), 16), dtype=torch.uint8, device=dev)
            _lib.call('sfvos_conv3d_wgrad', ctypes.byref(d), _ptr(acts[i - 1]), _ptr(cur), _ptr(g_w), 0, _ptr(wws),
                      _stream())
            conv_grads.append((g_w, g_b))
            if i > 1 or need[1]:
                dd = _conv_desc(dt_id, N, H, W, cout, cin, 0)
                nxt = torch.empty((N, H, W, cin), dtype=tdt, device=dev)
                _lib.call('sfvos_conv3d', ctypes.byref(dd), _ptr(cur), _ptr(head._packed(conv, dt_id, tdt, dgrad=True)),
                          None, _ptr(nxt), None, _stream())
                cur = nxt
        dx = _to_nchw(cur, dt_id) if need[1] else None
        for g_w, g_b in reversed(conv_grads):
            grads += [g_w, g_b]
        grads += tail
        return (None, dx) + tuple(g if n else None for g, n in zip(grads, need[2:]))


class _MaskLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, targets):
        N, K = logits.shape[0], logits.shape[1]
        P = logits[0, 0].numel()
        lg = logits.detach().float().contiguous()
        lab = labels.to(device=lg.device, dtype=torch.int64).contiguous()
        tg = targets.detach().to(device=lg.device, dtype=torch.float32).contiguous()
        if lab.numel() != N or tg.numel() != N * P:
            raise RuntimeError('maskrcnn_loss: one label and one [M,M] target per RoI expected')
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        _lib.call('sfvos_mask_bce_loss', _ptr(lg), _ptr(lab), _ptr(tg), N, K, P, _ptr(loss), _stream())
        ctx.save_for_backward(lg, lab, tg)
        return loss

    @staticmethod
    def backward(ctx, up):
        lg, lab, tg = ctx.saved_tensors
        N, K = lg.shape[0], lg.shape[1]
        P = lg[0, 0].numel()
        upc = up.detach().float().contiguous()
        d = torch.empty_like(lg)
        _lib.call('sfvos_mask_bce_loss_grad', _ptr(lg), _ptr(lab), _ptr(tg), _ptr(upc), N, K, P, _ptr(d), _stream())
        return d, None, None


def maskrcnn_loss(mask_logits, labels, mask_targets):
    """torchvision roi_heads.maskrcnn_loss behind `project_masks_on_boxes` (the RoIAlign of the ground-truth masks is
    torchvision's): mean binary cross-entropy with logits between the label's channel of `mask_logits` [N,K,M,M] and
    `mask_targets` [N,M,M]; `mask_logits.sum() * 0` when there is no positive RoI."""
    if mask_logits.shape[0] == 0 or mask_targets.numel() == 0:
        return mask_logits.sum() * 0
    _check_gpu(mask_logits, 'maskrcnn_loss')
    return _MaskLossFn.apply(mask_logits, labels, mask_targets)


class MaskBranch(nn.Module):
    """mask_head + mask_predictor as the reference's roi_heads holds them (attribute names = torchvision's, so
    `load_state_dict` takes the `roi_heads.` slice of a reference checkpoint)."""

    def __init__(self, in_channels=256, num_classes=2, precision='fp32'):
        super(MaskBranch, self).__init__()
        self.mask_head = MaskRCNNHeads(in_channels, (256, 256, 256, 256), 1, precision)
        self.mask_predictor = MaskRCNNPredictor(256, 256, num_classes, precision)   # model.py:20-25

    def _ordered_params(self):
        ps = []
        for i in range(1, self.mask_head.n_layers + 1):
            conv = getattr(self.mask_head, 'mask_fcn%d' % i)
            ps += [conv.weight, conv.bias]
        ps += [self.mask_predictor.conv5_mask.weight, self.mask_predictor.conv5_mask.bias,
               self.mask_predictor.mask_fcn_logits.weight, self.mask_predictor.mask_fcn_logits.bias]
        return ps

    def forward(self, roi_features):
        """RoIAligned features [N,256,14,14] -> mask logits [N,num_classes,28,28] (training-side output);
        differentiable w.r.t. the features and the parameters (backward on libsfvos kernels)."""
        ps = self._ordered_params()
        if roi_features.shape[0] == 0:
            # no positive proposal / no detection: torchvision's heads return an empty [0,K,2H,2W]; stay on the graph
            K = self.mask_predictor.mask_fcn_logits.out_channels
            out = roi_features.new_zeros((0, K, 2 * roi_features.shape[2], 2 * roi_features.shape[3]))
            if torch.is_grad_enabled() and roi_features.requires_grad:
                out = out + roi_features.sum() * 0
            return out
        if torch.is_grad_enabled() and \
                (roi_features.requires_grad or any(p.requires_grad for p in ps)):
            _check_gpu(roi_features, 'MaskBranch')
            return _MaskBranchFn.apply(self, roi_features, *ps)
        return self.mask_predictor.forward_from_nhwc(self.mask_head.forward_nhwc(roi_features))[0]

    @torch.no_grad()
    def predict(self, roi_features, labels, boxes, img_shape):
        """Inference path of roi_heads + transform.postprocess for one image (model.py:346-347): the detections'
        masks pasted into the image, [N,1,H,W] fp32 probabilities (thresholded at 0.5 and OR-ed by
        davis_evaluate.py:40-42 -> sfvos_amd.union_mask)."""
        if roi_features.shape[0] == 0:
            return roi_features.new_empty((0, 1, int(img_shape[0]), int(img_shape[1])))
        _, prob = self.mask_predictor.forward_from_nhwc(self.mask_head.forward_nhwc(roi_features), labels,
                                                        want_logits=False, want_prob=True)
        return paste_masks_in_image(prob, boxes, img_shape)
