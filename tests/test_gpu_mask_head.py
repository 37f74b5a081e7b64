"""GPU: the mask branch behind RoIAlign (MaskRCNNHeads + MaskRCNNPredictor + maskrcnn_inference +
paste_masks_in_image, reference code/helpers/model.py:17-25,346-347) on libsfvos kernels against the torch-core
restatement in oracle/mask_head_ref.py.

PARITY UNPINNED BY THE REFERENCE: the arithmetic is torchvision's (third-party, not vendored, not installed here) and no
reference fixture covers it; the oracle restates torchvision's published modules.  Tolerances: fp32 1e-4 of the tensor
scale (exact-f32 MFMA, summation order only), bf16 2e-2; pasted probabilities 1e-5 absolute; the >= 0.5 union
(davis_evaluate.py:40-42) identical wherever the oracle's probability is not within 1e-5 of 0.5."""
import numpy as np
import pytest
import torch

from oracle.mask_head_ref import OracleMaskBranch
from oracle.mask_head_ref import maskrcnn_inference as ref_inference
from oracle.mask_head_ref import paste_masks_in_image as ref_paste

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def relmax(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def make(precision):
    from sfvos_amd import MaskBranch
    torch.manual_seed(5)
    o = OracleMaskBranch(256, 2)
    with torch.no_grad():
        for p in o.parameters():   # non-zero biases, so that they are exercised
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape) * 0.1)
    m = MaskBranch(256, 2, precision)
    m.load_state_dict(o.state_dict(), strict=True)
    return m.to(DEV), o


BOXES = torch.tensor([[10.3, 20.7, 110.2, 90.5],      # ordinary
                      [-15.0, -8.0, 40.0, 30.0],      # sticks out at the top left
                      [800.0, 400.0, 870.0, 500.0],   # sticks out at the bottom right
                      [200.0, 100.0, 201.0, 101.5],   # tiny
                      [300.5, 50.2, 300.5, 50.2],     # degenerate (zero area)
                      [0.0, 0.0, 853.0, 479.0],       # the whole image
                      [420.9, 250.1, 470.3, 470.8]])  # tall


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_mask_branch_logits_match_the_oracle(precision):
    m, o = make(precision)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(7, 256, 14, 14, generator=g).relu()      # RoIAlign output of post-ReLU-ish features
    if precision == 'bf16':
        x = x.bfloat16().float()
    with torch.no_grad():
        ref = o(x)
        got = m(x.to(DEV))
        h_ref = o.mask_head(x)
        h_got = m.mask_head(x.to(DEV))
    assert tuple(got.shape) == (7, 2, 28, 28) and got.dtype == torch.float32
    tol = 1e-4 if precision == 'fp32' else 2e-2
    e_h, e_l = relmax(h_got.cpu(), h_ref), relmax(got.cpu(), ref)
    print('mask branch %s: heads max err / scale %.2e, logits %.2e' % (precision, e_h, e_l))
    assert e_h < tol and e_l < tol
    # the predictor alone, through its NCHW entry point
    with torch.no_grad():
        p_ref = o.mask_predictor(h_ref)
        p_got = m.mask_predictor(h_ref.to(DEV))
    assert relmax(p_got.cpu(), p_ref) < tol


def test_inference_and_paste_match_the_oracle():
    from sfvos_amd import maskrcnn_inference, paste_masks_in_image, union_mask
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(7, 2, 28, 28, generator=g) * 3
    labels = torch.tensor([1, 1, 0, 1, 1, 1, 0])
    ref_p = ref_inference(logits, labels)
    got_p = maskrcnn_inference(logits.to(DEV), labels.to(DEV))
    assert tuple(got_p.shape) == (7, 1, 28, 28)
    assert float((got_p.cpu() - ref_p).abs().max()) < 1e-6
    H, W = 480, 854
    ref = ref_paste(ref_p, BOXES, (H, W))
    got = paste_masks_in_image(ref_p.to(DEV), BOXES.to(DEV), (H, W))
    assert tuple(got.shape) == (7, 1, H, W)
    err = (got.cpu() - ref).abs()
    for i in range(7):
        print('paste box %d: max abs err %.2e, pasted pixels %d' % (i, float(err[i].max()), int((ref[i] != 0).sum())))
    assert float(err.max()) < 1e-5
    # the evaluation-side reducer on top (davis_evaluate.py:40-42): OR_k (mask_k >= 0.5)
    total = np.zeros((H, W), dtype=bool)
    for mk in ref:
        total = np.logical_or(total, (mk.numpy() >= 0.5)[0])
    uni = union_mask(got).cpu().numpy()
    near = ((ref - 0.5).abs() < 1e-5).any(0)[0].numpy()
    assert np.array_equal(uni[~near], total[~near])
    assert paste_masks_in_image(ref_p[:0].to(DEV), BOXES[:0].to(DEV), (H, W)).shape == (0, 1, H, W)


def test_predict_end_to_end_matches_the_oracle():
    m, o = make('fp32')
    g = torch.Generator().manual_seed(21)
    x = torch.randn(7, 256, 14, 14, generator=g).relu()
    labels = torch.tensor([1, 1, 1, 1, 1, 1, 1])      # the reference has one foreground class (num_classes = 2)
    H, W = 480, 854
    with torch.no_grad():
        ref = ref_paste(ref_inference(o(x), labels), BOXES, (H, W))
    got = m.predict(x.to(DEV), labels.to(DEV), BOXES.to(DEV), (H, W))
    assert float((got.cpu() - ref).abs().max()) < 1e-4


def test_mask_head_refuses_cpu():
    from sfvos_amd import MaskBranch, paste_masks_in_image
    m = MaskBranch()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 256, 14, 14))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        paste_masks_in_image(torch.zeros(1, 1, 28, 28), torch.zeros(1, 4), (8, 8))
