"""GPU: sfvos_amd.SlowFastLayers (HIP path, through the C ABI) against
  (a) the golden fixtures generated from the reference's own class (tests/golden, oracle/make_golden.py),
  (b) the CPU oracle on seeded inputs, and
  (c) size-independent properties at BASELINE.json's full sizes.

Tolerances (BASELINE.json north_star): fp32 logits within 1e-3 relative -- measured against the
tensor's own scale (max |a-b| / max |b|); the discrete per-pixel argmax proxy bit-exact in fp32
mode; bf16 error is measured and bounded separately (5e-2, stated in the test)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from golden_util import (BIG_LEVELS, CONFIGS, SMALL_LEVELS, clip_inputs, load_case, max_rel_err, rel_err, sample_idx)
from oracle.closed_form import closed_form_features, closed_form_state_dict
from oracle.slowfast_ref import OracleSlowFastLayers, proxy_loss

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3


def build(sp, fp, precision='fp32'):
    from sfvos_amd import SlowFastLayers
    dev = torch.device('cuda:0')
    m = SlowFastLayers(256, dev, sp, fp, precision=precision)
    m.load_state_dict(closed_form_state_dict(m))
    return m.to(dev), dev


def _check_against_fixture(sp, fp, levels, g, use_fused_sgd):
    from sfvos_amd import FusedSGD
    m, dev = build(sp, fp, 'fp32')
    # ---- eval mode
    m.eval()
    with torch.no_grad():
        slow, fast = clip_inputs(sp, fp, levels, 0, dev, clips=g['clips'])
        out = m.temporally_enhance_features(slow, fast)
    assert list(out.keys()) == list(levels.keys())
    for k, v in out.items():
        ref = g['eval_out/%s' % k]
        assert tuple(v.shape) == ref.shape and v.dtype == torch.float32
        assert max_rel_err(v.cpu().numpy(), ref) < FP32_TOL
        assert np.array_equal(v.argmax(1).cpu().numpy(), torch.from_numpy(ref).argmax(1).numpy()), \
            'argmax proxy differs (eval, level %s)' % k
    # ---- two accumulated training clips + one SGD step (model.py:369-374, train.py:80)
    m.train()
    params = list(m.parameters())
    opt = FusedSGD(params, lr=1e-3, momentum=0.9, weight_decay=1e-4).attach(m) if use_fused_sgd else \
        torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=1e-4)
    opt.zero_grad()
    for clip in (0, 1):
        slow, fast = clip_inputs(sp, fp, levels, clip, dev, clips=g['clips'])
        out = m.temporally_enhance_features(slow, fast)
        loss = proxy_loss(out)
        loss.backward()
        ref_loss = float(g['loss/%d' % clip])
        assert abs(loss.item() - ref_loss) < FP32_TOL * abs(ref_loss)
        if clip == 0:
            for k, v in out.items():
                ref = g['out/0/%s' % k]
                assert max_rel_err(v.detach().cpu().numpy(), ref) < FP32_TOL
                assert np.array_equal(v.argmax(1).cpu().numpy(), torch.from_numpy(ref).argmax(1).numpy()), \
                    'argmax proxy differs (train, level %s)' % k
    gscale = max(float(g['gnorm/%s' % k]) for k, _ in m.named_parameters())
    for key, p in m.named_parameters():
        gr = p.grad.detach().reshape(-1).cpu()
        ref_n = float(g['gnorm/%s' % key])
        ref_s = g['gsamp/%s' % key]
        if key.endswith('conv1.bias') or key.endswith('conv2.bias') or key.endswith('conv3.bias'):
            # conv bias feeding a train-mode BN: the true gradient is 0, both sides hold round-off
            assert float(gr.double().norm()) < 1e-4 * gscale
            continue
        assert abs(float(gr.double().norm()) - ref_n) <= FP32_TOL * ref_n + 1e-7, key
        assert np.abs(gr[sample_idx(gr.numel())].numpy() - ref_s).max() <= FP32_TOL * np.abs(ref_s).max() + 1e-7, key
    for key, b in m.named_buffers():
        ref = g['stat/%s' % key]
        if key.endswith('num_batches_tracked'):
            assert int(b) == int(ref)
        else:
            assert max_rel_err(b.cpu().numpy(), ref) < FP32_TOL, key
    opt.step()
    for key, p in m.named_parameters():
        v = p.detach().reshape(-1).cpu()
        assert abs(float(v.double().norm()) - float(g['pnorm/%s' % key])) <= 1e-5 * float(g['pnorm/%s' % key]), key
        assert np.abs(v[sample_idx(v.numel())].numpy() - g['psamp/%s' % key]).max() < 1e-5, key


@pytest.mark.parametrize('sp,fp', CONFIGS)
def test_module_matches_reference_fixture_small(sp, fp):
    _check_against_fixture(sp, fp, SMALL_LEVELS, load_case(sp, fp, 'small'), use_fused_sgd=(fp % 2 == 1))


def test_module_matches_reference_fixture_big():
    _check_against_fixture(3, 7, BIG_LEVELS, load_case(3, 7, 'big'), use_fused_sgd=True)


def test_input_gradient_matches_reference_fixture():
    """OSVOS with a trainable backbone needs dgrad into the features (osvos_model.py:50,64)."""
    g = load_case(3, 7, 'inputgrad')
    m, dev = build(3, 7, 'fp32')
    m.train()
    from oracle.closed_form import closed_form_features, slice_slow
    fast = closed_form_features(7, SMALL_LEVELS, clip=int(g['clips'][0]))
    fast = OrderedDict((k, v.to(dev).requires_grad_(True)) for k, v in fast.items())
    out = m.temporally_enhance_features([slice_slow(fast, 3)], [fast])
    proxy_loss(out).backward()
    for k, v in fast.items():
        gr = v.grad.reshape(-1).cpu()
        assert abs(float(gr.double().norm()) - float(g['ignorm/%s' % k])) <= FP32_TOL * float(g['ignorm/%s' % k])
        ref = g['igsamp/%s' % k]
        assert np.abs(gr[sample_idx(gr.numel())].numpy() - ref).max() <= FP32_TOL * np.abs(ref).max() + 1e-8


def test_forward_method_and_eval_train_difference():
    m, dev = build(3, 7, 'fp32')
    o = OracleSlowFastLayers(256, torch.device('cpu'), 3, 7)
    o.load_state_dict(closed_form_state_dict(o))
    g = torch.Generator().manual_seed(63)
    fast = torch.randn(2, 256, 7, 9, 14, generator=g)
    slow = fast[:, :, 2:5]
    for mode in ('train', 'eval'):
        getattr(m, mode)()
        getattr(o, mode)()
        with torch.no_grad():
            s, f = m(slow.to(dev), fast.to(dev))
            rs, rf = o(slow, fast)
        assert tuple(s.shape) == (2, 224, 1, 9, 14) and tuple(f.shape) == (2, 32, 1, 9, 14)
        assert max_rel_err(s.cpu().numpy(), rs.numpy()) < FP32_TOL
        assert max_rel_err(f.cpu().numpy(), rf.numpy()) < FP32_TOL


def test_state_dict_interchange_with_reference_format():
    """Checkpoints written by either side load on the other with strict=True (train.py:90,115-117)."""
    m, dev = build(1, 7, 'fp32')
    o = OracleSlowFastLayers(256, torch.device('cpu'), 1, 7)
    sd = m.state_dict()
    assert list(sd.keys()) == list(o.state_dict().keys())
    o.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)
    m.load_state_dict(o.state_dict(), strict=True)


def test_frozen_parameters_and_no_grad():
    m, dev = build(3, 3, 'fp32')
    m.train()
    for p in m.slow_conv1.parameters():
        p.requires_grad = False
    slow, fast = clip_inputs(3, 3, SMALL_LEVELS, 0, dev)
    out = m.temporally_enhance_features(slow, fast)
    proxy_loss(out).backward()
    assert m.slow_conv1.weight.grad is None and m.fast_conv1.weight.grad is not None
    with torch.no_grad():
        out = m.temporally_enhance_features(slow, fast)
    assert not out['0'].requires_grad


@pytest.mark.parametrize('freeze', [None, 'bn_s2.weight'])
def test_gradient_sink_matches_autograd_accumulation(freeze):
    """FusedSGD.attach(): gradients written / accumulated straight into the flat buffer by the kernels equal
    the ones autograd accumulates from temporaries, over two clips (overwrite, then accumulate), also when a
    BatchNorm has only one of its two parameters in the optimiser."""
    from sfvos_amd import FusedSGD
    flats = []
    for attach in (True, False):
        m, dev = build(3, 7, 'fp32')
        m.train()
        if freeze:
            mod, attr = freeze.split('.')
            getattr(m, mod)._parameters[attr].requires_grad = False
        opt = FusedSGD(m.parameters())
        if attach:
            opt.attach(m)
        opt.zero_grad()
        for clip in (0, 1):
            slow, fast = clip_inputs(3, 7, SMALL_LEVELS, clip, dev)
            proxy_loss(m.temporally_enhance_features(slow, fast)).backward()
        flats.append(opt.flat_grad.clone())
        if attach:  # the sink must have been used: autograd saw no gradient for the conv weights
            assert m._grad_sink is opt and not opt._clean
    scale = float(flats[1].abs().max())
    assert scale > 0 and float((flats[0] - flats[1]).abs().max()) <= 1e-6 * scale


def test_cpu_tensors_are_refused():
    from sfvos_amd import SlowFastLayers
    m = SlowFastLayers(256, torch.device('cpu'), 1, 1)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 256, 1, 4, 4), torch.zeros(1, 256, 1, 4, 4))


@pytest.mark.parametrize('sp,fp', [(3, 7), (4, 32), (4, 64)])
def test_bf16_path_error_is_bounded(sp, fp):
    """bf16 storage / f32 accumulate (the dtype the headline number is quoted in), forward AND backward against the
    fp32 fixtures generated from the reference's class: errors are measured and printed; bounds (stated here): fused
    maps 5e-2 of the output scale, loss 2e-2, every parameter gradient within 5e-2 of its scale (norm and the 64 samples;
    measured 0.5-2e-2).  Argmax agreement is reported, not required."""
    g = load_case(sp, fp, 'small')
    m, dev = build(sp, fp, 'bf16')
    m.train()
    slow, fast = clip_inputs(sp, fp, SMALL_LEVELS, 0, dev, clips=g['clips'])
    out = m.temporally_enhance_features(slow, fast)
    loss = proxy_loss(out)
    loss.backward()
    for k, v in out.items():
        ref = g['out/0/%s' % k]
        e = max_rel_err(v.detach().cpu().numpy(), ref)
        agree = float((v.argmax(1).cpu().numpy() == torch.from_numpy(ref).argmax(1).numpy()).mean())
        print('bf16 (%d,%d) level %s: max err / scale = %.3e, rel-L2 = %.3e, argmax agreement = %.4f'
              % (sp, fp, k, e, rel_err(v.detach().cpu().numpy(), ref), agree))
        assert e < 5e-2
    assert abs(loss.item() - float(g['loss/0'])) < 2e-2 * abs(float(g['loss/0']))
    # gradients of the first clip alone: the fixture holds them for the sum of two clips, so run the second one too
    slow, fast = clip_inputs(sp, fp, SMALL_LEVELS, 1, dev, clips=g['clips'])
    proxy_loss(m.temporally_enhance_features(slow, fast)).backward()
    gscale = max(float(g['gnorm/%s' % k]) for k, _ in m.named_parameters())
    rows = []
    for key, p in m.named_parameters():
        gr = p.grad.detach().reshape(-1).cpu()
        ref_n, ref_s = float(g['gnorm/%s' % key]), g['gsamp/%s' % key]
        if key.endswith('conv1.bias') or key.endswith('conv2.bias') or key.endswith('conv3.bias'):
            assert float(gr.double().norm()) < 2e-2 * gscale   # true gradient 0 (bias in front of train-mode BN)
            continue
        got_s = gr[sample_idx(gr.numel())].numpy()
        rows.append((key, abs(float(gr.double().norm()) - ref_n) / ref_n, rel_err(got_s, ref_s),
                     float(np.abs(got_s - ref_s).max() / np.abs(ref_s).max())))
    for key, en, el2, emax in rows:
        print('bf16 (%d,%d) grad %-18s norm err %.2e, samples rel-L2 %.2e, max / scale %.2e' % (sp, fp, key, en, el2, emax))
    # bounds (stated): gradient norms within 5e-2; the 64 sampled entries within 0.15 rel-L2 (the first-layer weight
    # gradients are heavily cancelling sums -- dx is orthogonal to 1 and to x-hat per channel -- which amplifies the
    # bf16 rounding of dx and x: measured 0.5-12e-2 per entry at 312 positions, 1e-3 on the norms)
    for key, en, el2, emax in rows:
        assert en < 5e-2 and el2 < 0.15, (key, en, el2, emax)


# ---- parity at the BENCHMARKED size, dtype and path (bench.py: (4,32), the 5-level DAVIS pyramid incl. level '0',
# PackedClip hand-over, FusedSGD.attach gradient sink, overwrite + accumulate) against the CPU oracle run on this
# box's host cores (model.py:118-165,369-374) -----------------------------------------------------------------
def _full_size_oracle(sp, fp, keys=None):
    import time
    from sfvos_amd import davis_pyramid
    from oracle.slowfast_ref import _target
    t_start = time.time()
    pyr = [(k, hw) for k, hw in davis_pyramid() if keys is None or k in keys]
    gen = torch.Generator().manual_seed(63)
    # bf16-representable values, so the fp32 and the bf16 run (and the oracle) see the very same clip
    fast = OrderedDict((k, torch.randn(fp, 256, h, w, generator=gen).bfloat16().float()) for k, (h, w) in pyr)
    idx = fp // 2
    slow = OrderedDict((k, v[idx - sp // 2: idx + (sp + 1) // 2]) for k, v in fast.items())
    o = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    o.load_state_dict(closed_form_state_dict(o))
    o.train()
    opt = torch.optim.SGD(o.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    out = o.temporally_enhance_features([slow], [fast])
    loss = proxy_loss(out)
    loss.backward()
    res = dict(sp=sp, fp=fp, pyr=pyr, fast=fast, slow=slow, loss=float(loss),
               out={k: v.detach().clone() for k, v in out.items()},
               grad={k: p.grad.detach().clone() for k, p in o.named_parameters()},
               target={k: _target(k, tuple(v.shape), v.device) for k, v in out.items()})
    del out, loss
    # the module under test runs the SAME clip twice (overwrite, then accumulate into the gradient sink, as bench.py's
    # two clips per optimiser step): gradients are exactly 2x, the running statistics see two updates per level
    with torch.no_grad():
        o.temporally_enhance_features([slow], [fast])
        for p in o.parameters():
            p.grad.mul_(2.0)
    res['stat'] = {k: b.detach().clone() for k, b in o.named_buffers()}
    opt.step()
    res['param'] = {k: p.detach().clone() for k, p in o.named_parameters()}
    print('[oracle (%d,%d) on levels %s: %.0f s on %d host threads]'
          % (sp, fp, ','.join(k for k, _ in pyr), time.time() - t_start, torch.get_num_threads()), flush=True)
    return res


@pytest.fixture(scope='module')
def full_size_oracle():
    return _full_size_oracle(4, 32)


@pytest.fixture(scope='module')
def full_size_oracle_c4():
    """BASELINE config 4: (sp, fp) = (4, 64) at DAVIS size -- fast convs with kt = 22, the laterals with
    kt = 41 / 21 (weight images at / beyond the LDS budget: the last tap in registers).  Levels '1', '2', '3', 'pool'
    (21 420 positions, 96x168 ... 12x21) by default: the CPU oracle needs several minutes for level '0' alone at 64
    frames (fp32 torch on the host); SFVOS_C4_ALL_LEVELS=1 adds it (64-frame level-'0' addressing: 2.8 GB bf16 clip)."""
    import os
    keys = None if os.environ.get('SFVOS_C4_ALL_LEVELS') else ('1', '2', '3', 'pool')
    return _full_size_oracle(4, 64, keys)


# Gates of the full-size comparison (each at most 2x what was measured on MI355X, see the prints of the tests):
#   out: fused maps, max |a-b| / max|b|;  g3: layer-3 gradients (no ReLU behind them), (rel-L2, max entry / scale);
#   g12: layer-1/2 gradients (behind ReLU masks: DESIGN.md section 2 and the mask-flip experiment below).
FULL_GATES = {
    ('c2', 'fp32'): dict(out=2e-5, loss=1e-5, g3=(1e-4, 1e-4), g12=(5e-3, 2e-2), stat=1e-4, param=1e-5),
    ('c2', 'bf16'): dict(out=5e-2, loss=2e-2, g3=(2e-2, 2e-2), g12=(0.15, 0.3), stat=2e-2, param=1e-3),
    # (4,64) on 21 420 positions: a flipped ReLU mask weighs 4x more than at 85 932 -- 70 flips move the layer-1/2
    # gradients by 1.3e-2 rel-L2 (single entries of slow_conv2.weight by 0.13 of the scale); with the SAME masks on both
    # sides they agree at 5e-6 (test_full_size_fp32_relu_mask_flips_explain_the_gradient_error[c4])
    ('c4', 'fp32'): dict(out=2e-5, loss=1e-5, g3=(1e-4, 1e-4), g12=(3e-2, 0.3), stat=1e-4, param=1e-4),
    ('c4', 'bf16'): dict(out=5e-2, loss=2e-2, g3=(2e-2, 2e-2), g12=(0.15, 0.3), stat=2e-2, param=1e-3),
}


def _check_full_size(r, precision, tag):
    from sfvos_amd import FusedSGD, MSEProxyLoss, PackedClip
    gates = FULL_GATES[(tag, precision)]
    sp, fp, pyr = r['sp'], r['fp'], r['pyr']
    m, dev = build(sp, fp, precision)
    m.train()
    opt = FusedSGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4).attach(m)
    tdt = torch.bfloat16 if precision == 'bf16' else torch.float32
    levels = [r['fast'][k].to(dev).permute(0, 2, 3, 1).unsqueeze(0).to(tdt).contiguous() for k, _ in pyr]
    clip = PackedClip.from_levels(levels, keys=[k for k, _ in pyr],
                                  layout='grouped' if precision == 'bf16' else 'ndhwc')
    del levels
    loss_fn = MSEProxyLoss({k: v.to(dev) for k, v in r['target'].items()})
    opt.zero_grad()
    out = m.enhance_packed(clip)
    loss = loss_fn(out)
    loss.backward()                      # overwrite mode of the gradient sink
    loss_fn(m.enhance_packed(clip)).backward()   # accumulate mode
    assert m._grad_sink is opt
    for k, v in out.items():
        ref = r['out'][k]
        got = v.detach().cpu()
        e = max_rel_err(got.numpy(), ref.numpy())
        top2 = ref.topk(2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1])
        scale = float(ref.abs().max())
        err_px = (got - ref).abs().amax(1)          # this implementation's deviation at each pixel
        same = got.argmax(1) == ref.argmax(1)
        print('%s (%d,%d) full size level %s: max err / scale %.3e, rel-L2 %.3e, argmax agreement %.6f (%d of %d pixels '
              'differ)' % (precision, sp, fp, k, e, rel_err(got.numpy(), ref.numpy()), float(same.float().mean()),
                           int((~same).sum()), same.numel()))
        assert e < gates['out'], k
        if precision == 'fp32':
            # Where the argmax differs, the oracle's top-2 margin must be below the deviation of THAT pixel (x2: both
            # channels move) -- i.e. the pixel is a near-tie that fp32 round-off decides -- and that deviation is itself
            # gated above (measured 4e-6 of the scale): no exemption threshold is assumed.
            for b_, h_, w_ in (~same).nonzero().tolist():
                mg, ep = float(margin[b_, h_, w_]), float(err_px[b_, h_, w_])
                print('   argmax differs at level %s pixel (%d,%d): oracle top-2 margin %.3e of scale, deviation there '
                      '%.3e of scale' % (k, h_, w_, mg / scale, ep / scale))
                assert mg <= 2.0 * ep and mg < gates['out'] * scale, (k, h_, w_, mg, ep)
            assert int((~same).sum()) <= 1e-4 * same.numel(), k
    assert abs(loss.item() - r['loss']) < gates['loss'] * abs(r['loss']), (loss.item(), r['loss'])
    gscale = max(float(v.abs().max()) for v in r['grad'].values())
    # gradients: rel-L2 and max-entry error of the whole tensors.
    # Layer 3 (no ReLU behind it): fp32 1e-4, bf16 2e-2 -- measured 1e-7..3e-6 and 1e-4..7e-3.
    # Layers 1-2: the gradient passes through ReLU masks.  Of the oracle's ~200 M ReLU inputs at this size, ~1e-5
    # (fp32) / ~1e-2 (bf16) lie within the other implementation's round-off of zero and get the opposite mask, which
    # moves sums over ~1 M positions by ~1e-3 (fp32, measured 0.3-1.6e-3 rel-L2) / ~7e-2 (bf16, measured 5-8e-2):
    # the error JUMPS between layer 3 and layer 2 and does not grow from layer 2 to layer 1 -- the signature of mask
    # flips, not of accumulated arithmetic error: test_full_size_fp32_relu_mask_flips_explain_the_gradient_error counts
    # the flipped masks and shows the gradients agree at 1e-4 once both sides use the same masks.
    (g_l2, g_max), (g3_l2, g3_max) = gates['g12'], gates['g3']
    rows = []
    for key, p in m.named_parameters():
        ref = 2.0 * r['grad'][key]
        got = p.grad.detach().cpu()
        if key.endswith('conv1.bias') or key.endswith('conv2.bias') or key.endswith('conv3.bias'):
            assert float(got.abs().max()) < (1e-4 if precision == 'fp32' else 2e-2) * 2 * gscale, key
            continue
        rows.append((key, rel_err(got.numpy(), ref.numpy()), max_rel_err(got.numpy(), ref.numpy())))
    for key, el2, emax in rows:
        print('%s (%d,%d) full size grad %-18s rel-L2 %.2e, max / scale %.2e' % (precision, sp, fp, key, el2, emax))
    for key, el2, emax in rows:
        layer3 = key.split('.')[0] in ('fast_conv3', 'slow_conv3', 'bn_f3', 'bn_s3')
        assert el2 < (g3_l2 if layer3 else g_l2) and emax < (g3_max if layer3 else g_max), (key, el2, emax)
    for key, b in m.named_buffers():
        ref = r['stat'][key]
        if key.endswith('num_batches_tracked'):
            assert int(b) == int(ref)
        else:
            assert max_rel_err(b.cpu().numpy(), ref.numpy()) < gates['stat'], key
    opt.step()
    for key, p in m.named_parameters():
        ref = r['param'][key]
        d = float((p.detach().cpu() - ref).abs().max())
        assert d <= gates['param'] * float(ref.abs().max()) + 1e-9, (key, d)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_full_size_headline_config_matches_the_oracle(full_size_oracle, precision):
    """The BENCHMARKED configuration (bench.py: (4,32), the 5-level DAVIS pyramid, PackedClip hand-over, FusedSGD.attach
    gradient sink, overwrite + accumulate) against the CPU oracle run on this box's host cores.
    fp32: fused maps, loss, BN running statistics, EVERY parameter gradient (whole tensors) and the parameters after
    the SGD step against the oracle (gates FULL_GATES: at most 2x the measured error); per-pixel argmax over the 256
    fused channels identical except at near-ties that fp32 round-off decides (each printed with its margin).
    bf16 (the bench dtype, channel-group-major clip): the same quantities, errors measured and printed, bounded at
    5e-2 (outputs, gradients), 2e-2 (loss); argmax agreement reported."""
    _check_full_size(full_size_oracle, precision, 'c2')


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_full_size_config4_matches_the_oracle(full_size_oracle_c4, precision):
    """BASELINE config 4 = (sp, fp) = (4, 64) at DAVIS size (levels: see the fixture), the same checks as the headline
    configuration: this is the only place the kt = 22 frame-split conv and the kt = 41 / 21 lateral kernels (weight image
    at the LDS budget, last tap in registers) run at DAVIS size against the oracle (Missing 5)."""
    _check_full_size(full_size_oracle_c4, precision, 'c4')


def _gpu_relu_masks(m, out, plan, pyr):
    """The ReLU masks the GPU module used in the forward that produced `out` (dict of its fused maps, still on the
    graph): the post-ReLU activations it stored for its backward, as {(level key, bn name): bool [1,C,T,H,W]}."""
    state = next(iter(out.values())).grad_fn.state
    masks = {}
    for l in plan.layers:
        if not l.relu:
            continue
        act = state.bufs[l.dst][:, l.dst_off: l.dst_off + l.c_out]
        off = 0
        for k, (h, w) in pyr:
            n = l.t_out * h * w
            a = act[off: off + n].reshape(1, l.t_out, h, w, l.c_out).permute(0, 4, 1, 2, 3)
            masks[(k, l.bn)] = (a > 0).cpu()
            off += n
    return masks


@pytest.mark.parametrize('tag', ['c2', 'c4'])
def test_full_size_fp32_relu_mask_flips_explain_the_gradient_error(request, tag):
    """VERDICT r2 (weak 1): the layer-1/2 gradients of the fp32 path differ from the oracle's by ~1e-3 rel-L2 at full
    size while layer 3 agrees at 1e-6.  DESIGN.md attributes this to ReLU masks: a pre-activation within fp32 round-off
    of zero lands on different sides in two correct implementations.  Evidence instead of argument:
      (1) count the elements whose mask differs (GPU module vs oracle), per layer;
      (2) re-run the ORACLE with the GPU module's masks (y * mask instead of relu(y): same forward values up to the
          ~1e-6 pre-activations that flipped) and compare the gradients again: they must agree like layer 3 does."""
    from sfvos_amd import MSEProxyLoss, PackedClip
    r = request.getfixturevalue('full_size_oracle' if tag == 'c2' else 'full_size_oracle_c4')
    sp, fp, pyr = r['sp'], r['fp'], r['pyr']
    m, dev = build(sp, fp, 'fp32')
    m.train()
    levels = [r['fast'][k].to(dev).permute(0, 2, 3, 1).unsqueeze(0).contiguous() for k, _ in pyr]
    clip = PackedClip.from_levels(levels, keys=[k for k, _ in pyr], layout='ndhwc')
    del levels
    loss_fn = MSEProxyLoss({k: v.to(dev) for k, v in r['target'].items()})
    out = m.enhance_packed(clip)
    masks = _gpu_relu_masks(m, out, m.plan, pyr)
    loss_fn(out).backward()
    torch.cuda.synchronize()
    o = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    o.load_state_dict(closed_form_state_dict(o))
    o.train()
    o.relu_masks = masks
    proxy_loss(o.temporally_enhance_features([r['slow']], [r['fast']])).backward()
    total = sum(v.numel() for v in masks.values())
    flips = sum(o.relu_flips.values())
    per_layer = {}
    for (k, bn), n in o.relu_flips.items():
        per_layer[bn] = per_layer.get(bn, 0) + n
    print('fp32 (%d,%d) full size: %d of %d ReLU masks differ between the GPU module and the oracle (%.2e): %s'
          % (sp, fp, flips, total, flips / total, ', '.join('%s %d' % kv for kv in sorted(per_layer.items()))))
    assert len(o.relu_flips) == len(masks)
    assert flips <= 1e-4 * total, 'more mask flips than fp32 round-off explains'
    worst_plain = worst_masked = 0.0
    ref_plain = r['grad']
    for key, p in m.named_parameters():
        if key.endswith('conv1.bias') or key.endswith('conv2.bias') or key.endswith('conv3.bias'):
            continue
        got = p.grad.detach().cpu().numpy()
        e_plain = rel_err(got, ref_plain[key].numpy())
        e_mask = rel_err(got, dict(o.named_parameters())[key].grad.numpy())
        layer3 = key.split('.')[0] in ('fast_conv3', 'slow_conv3', 'bn_f3', 'bn_s3')
        print('fp32 (%d,%d) full size grad %-18s rel-L2 vs oracle %.2e, vs oracle with the SAME masks %.2e'
              % (sp, fp, key, e_plain, e_mask))
        if not layer3:
            worst_plain, worst_masked = max(worst_plain, e_plain), max(worst_masked, e_mask)
        assert e_mask < 1e-4, (key, e_mask)
    print('fp32 (%d,%d) full size: layer-1/2 gradients: worst rel-L2 %.2e against the oracle, %.2e once both sides use the '
          'same ReLU masks' % (sp, fp, worst_plain, worst_masked))
    if flips > 0:
        assert worst_masked < 0.2 * worst_plain, 'the mask flips do not explain the gradient error'


def _oracle_clip(sp, fp, fast_cpu, train=True):
    """CPU oracle on one clip (dict level -> [fp,256,H,W] fp32): fused maps, loss, parameter gradients."""
    from oracle.closed_form import slice_slow
    o = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    o.load_state_dict(closed_form_state_dict(o))
    o.train(train)
    out = o.temporally_enhance_features([slice_slow(fast_cpu, sp)], [fast_cpu])
    loss = proxy_loss(out)
    loss.backward()
    return ({k: v.detach() for k, v in out.items()}, float(loss),
            {k: p.grad.detach().clone() for k, p in o.named_parameters()})


def _assert_grads_match(m, ref_g, tol, what, tol_max=None):
    """rel-L2 of every parameter gradient below tol (and, when given, every entry within tol_max of the scale)."""
    gscale = max(float(v.abs().max()) for v in ref_g.values())
    for key, p in m.named_parameters():
        if key.endswith('conv1.bias') or key.endswith('conv2.bias') or key.endswith('conv3.bias'):
            assert float(p.grad.abs().max()) < max(tol, 1e-4) * gscale, (what, key)   # true gradient 0
            continue
        e = rel_err(p.grad.cpu().numpy(), ref_g[key].numpy())
        assert e < tol, (what, key, e)
        if tol_max is not None:
            assert max_rel_err(p.grad.cpu().numpy(), ref_g[key].numpy()) < tol_max, (what, key)


def test_packed_clip_matches_the_oracle_and_frame_lists():
    """enhance_packed (channels-last hand-over) against the CPU oracle on the same clip -- fused maps,
    loss and every parameter gradient, fp32 at 1e-3 -- and, as an extra, bit-identical to temporally_enhance_features
    on the same data."""
    from sfvos_amd import PackedClip
    sp, fp = 3, 7
    m, dev = build(sp, fp, 'fp32')
    m.train()
    slow, fast = clip_inputs(sp, fp, SMALL_LEVELS, 0, dev)
    ref_out, ref_loss, ref_g = _oracle_clip(sp, fp, OrderedDict((k, v.cpu()) for k, v in fast[0].items()))
    out = m.temporally_enhance_features(slow, fast)
    proxy_loss(out).backward()
    frames_g = {k: p.grad.clone() for k, p in m.named_parameters()}
    m2, _ = build(sp, fp, 'fp32')
    m2.train()
    levels = [fast[0][k].permute(0, 2, 3, 1).unsqueeze(0).contiguous() for k in SMALL_LEVELS]   # [1,T,H,W,C]
    clip = PackedClip.from_levels(levels, keys=list(SMALL_LEVELS.keys()))
    out2 = m2.enhance_packed(clip)
    loss2 = proxy_loss(out2)
    loss2.backward()
    for k in out2:
        assert max_rel_err(out2[k].detach().cpu().numpy(), ref_out[k].numpy()) < FP32_TOL, k
        assert torch.equal(out2[k].argmax(1).cpu(), ref_out[k].argmax(1)), k
        assert torch.equal(out[k], out2[k])
    assert abs(loss2.item() - ref_loss) < FP32_TOL * abs(ref_loss)
    _assert_grads_match(m2, ref_g, FP32_TOL, 'packed', FP32_TOL)
    for k, p in m2.named_parameters():
        assert torch.equal(p.grad, frames_g[k]), k


@pytest.mark.parametrize('precision,pad', [('fp32', (2, 0)), ('fp32', (0, 3)), ('bf16', (1, 2)), ('bf16', (3, 0))])
def test_packed_clip_zero_frames_by_pointer_match_the_oracle(precision, pad):
    """The reference pads windows that stick out of the sequence with zero FEATURE frames (model.py:215-225).  A
    PackedClip stores only the real frames and names the padding (pad=(before, after)): the padding frames have no
    storage at all -- in the level-major buffer the positions "before" a level's first frame belong to the previous
    level -- so any read of them would show up as a wrong result.  Checked against the CPU oracle fed with explicit zero
    frames: fused maps, loss, parameter gradients (train mode), both dtypes / layouts, and the gradient w.r.t. the
    stored frames comes back in the clip's own shape."""
    from sfvos_amd import PackedClip
    sp, fp = 3, 7
    m, dev = build(sp, fp, precision)
    m.train()
    n_real = fp - pad[0] - pad[1]

    def make(clip_id):
        real = closed_form_features(n_real, SMALL_LEVELS, clip=clip_id)
        if precision == 'bf16':
            real = OrderedDict((k, v.bfloat16().float()) for k, v in real.items())
        window = OrderedDict((k, torch.cat([torch.zeros(pad[0], *v.shape[1:]), v, torch.zeros(pad[1], *v.shape[1:])]))
                             for k, v in real.items())
        return real, window
    # like oracle/make_golden.py: among 24 closed-form clips take the one whose ReLU inputs stay farthest from zero
    # in the oracle, so that no mask can flip between two correct fp32 implementations
    from oracle.closed_form import slice_slow
    o = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    o.load_state_dict(closed_form_state_dict(o))
    o.train()
    best = (-1.0, 0)
    with torch.no_grad():
        for cid in range(24):
            _, w_ = make(cid)
            o.relu_margins = []
            o.temporally_enhance_features([slice_slow(w_, sp)], [w_])
            best = max(best, (min(o.relu_margins), cid))
    real, window = make(best[1])
    ref_out, ref_loss, ref_g = _oracle_clip(sp, fp, window)
    tdt = torch.bfloat16 if precision == 'bf16' else torch.float32
    levels = [real[k].to(dev).permute(0, 2, 3, 1).unsqueeze(0).to(tdt).contiguous() for k in SMALL_LEVELS]
    clip = PackedClip.from_levels(levels, keys=list(SMALL_LEVELS.keys()),
                                  layout='grouped' if precision == 'bf16' else 'ndhwc', pad=pad)
    assert clip.frames == n_real and clip.window == fp
    clip.data.requires_grad_(True)
    out = m.enhance_packed(clip)
    loss = proxy_loss(out)
    loss.backward()
    tol = FP32_TOL if precision == 'fp32' else 5e-2
    for k in out:
        assert max_rel_err(out[k].detach().cpu().numpy(), ref_out[k].numpy()) < tol, k
    assert abs(loss.item() - ref_loss) < (FP32_TOL if precision == 'fp32' else 2e-2) * abs(ref_loss)
    # gradients against the oracle: fp32 1e-3 rel-L2 (ReLU margin of the chosen clip %.1e); bf16 0.15 (bf16 activations
    # flip ~1 %% of the ReLU masks: see test_full_size_headline_config_matches_the_oracle)
    print('zero-frame clip %d: min |ReLU input| in the oracle %.2e' % (best[1], best[0]))
    _assert_grads_match(m, ref_g, FP32_TOL if precision == 'fp32' else 0.15, 'padded clip')
    # ... and BIT-IDENTICAL to the same module fed the window with materialised zero frames: a zero frame produced by
    # an empty buffer descriptor and a stored zero frame put the same bytes into LDS
    m2, _ = build(sp, fp, precision)
    m2.train()
    full = [window[k].to(dev).permute(0, 2, 3, 1).unsqueeze(0).to(tdt).contiguous() for k in SMALL_LEVELS]
    clip2 = PackedClip.from_levels(full, keys=list(SMALL_LEVELS.keys()), layout='grouped' if precision == 'bf16' else 'ndhwc')
    clip2.data.requires_grad_(True)
    out2 = m2.enhance_packed(clip2)
    proxy_loss(out2).backward()
    for k in out:
        assert torch.equal(out[k], out2[k]), k
    for (k, p), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(p.grad, p2.grad), k
    g_full = clip2.data.grad if precision == 'fp32' else clip2.data.grad.permute(1, 0, 2).reshape(clip2.data.shape[1], -1)
    g_part = clip.data.grad if precision == 'fp32' else clip.data.grad.permute(1, 0, 2).reshape(clip.data.shape[1], -1)
    off_f = off_p = 0
    for (H, W) in SMALL_LEVELS.values():   # the stored frames' gradient = those frames of the full window's gradient
        a = g_full[off_f: off_f + fp * H * W].view(fp, H * W, -1)[pad[0]: pad[0] + n_real]
        b = g_part[off_p: off_p + n_real * H * W].view(n_real, H * W, -1)
        assert torch.equal(a, b)
        off_f += fp * H * W
        off_p += n_real * H * W
    assert clip.data.grad is not None and clip.data.grad.shape == clip.data.shape
    assert bool(torch.isfinite(clip.data.grad.float()).all()) and float(clip.data.grad.float().abs().max()) > 0
    with pytest.raises(RuntimeError):
        m.enhance_packed(PackedClip.from_levels(levels, keys=list(SMALL_LEVELS.keys()),
                                                layout='grouped' if precision == 'bf16' else 'ndhwc', pad=(pad[0] + 1, pad[1])))


def test_grouped_packed_clip_matches_ndhwc_and_returns_grouped_input_gradient():
    """bf16: a clip handed over channel-group-major (PackedClip layout 'grouped') gives the same bits as the same
    clip in pyramid NDHWC -- outputs, parameter gradients and the gradient w.r.t. the clip, which comes back in
    the layout the clip was given in; temporally_enhance_features (which builds the grouped layout itself in bf16)
    agrees with both."""
    from sfvos_amd import PackedClip
    sp, fp = 3, 7
    _, fast = clip_inputs(sp, fp, SMALL_LEVELS, 0, torch.device('cuda:0'))
    levels = [fast[0][k].permute(0, 2, 3, 1).unsqueeze(0).contiguous().bfloat16() for k in SMALL_LEVELS]  # [1,T,H,W,C]
    res = {}
    for layout in ('ndhwc', 'grouped'):
        m, dev = build(sp, fp, 'bf16')
        m.train()
        clip = PackedClip.from_levels(levels, keys=list(SMALL_LEVELS.keys()), layout=layout)
        assert clip.layout == layout and clip.channels == 256
        clip.data.requires_grad_(True)
        out = m.enhance_packed(clip)
        proxy_loss(out).backward()
        gx = clip.data.grad
        assert gx.shape == clip.data.shape
        if layout == 'grouped':
            gx = gx.permute(1, 0, 2).reshape(gx.shape[1], -1)
        res[layout] = (out, {k: p.grad.clone() for k, p in m.named_parameters()}, gx)
    for k in res['ndhwc'][0]:
        assert torch.equal(res['ndhwc'][0][k], res['grouped'][0][k]), k
    for k in res['ndhwc'][1]:
        assert torch.equal(res['ndhwc'][1][k], res['grouped'][1][k]), k
    assert torch.equal(res['ndhwc'][2], res['grouped'][2])
    m, dev = build(sp, fp, 'bf16')
    m.train()
    fast_bf = [OrderedDict((k, v.bfloat16().float()) for k, v in fast[0].items())]
    from oracle.closed_form import slice_slow
    slow_bf = [slice_slow(fast_bf[0], sp)]
    out3 = m.temporally_enhance_features(slow_bf, fast_bf)
    for k in out3:
        assert torch.equal(out3[k], res['grouped'][0][k]), k


def test_optimizer_checkpoint_interchange_with_torch_sgd():
    """FusedSGD.state_dict() / load_state_dict() use torch.optim.SGD's layout (reference train.py:117-121 saves it):
    a run checkpointed by either optimiser resumes on the other with identical parameters afterwards."""
    from sfvos_amd import FusedSGD

    def grads(m, clip):
        slow, fast = clip_inputs(1, 7, SMALL_LEVELS, clip, torch.device('cuda:0'))
        proxy_loss(m.temporally_enhance_features(slow, fast)).backward()

    def run(first, second):
        m, dev = build(1, 7, 'fp32')
        m.train()
        make = {'torch': lambda ps: torch.optim.SGD(ps, lr=1e-2, momentum=0.9, weight_decay=1e-4),
                'fused': lambda ps: FusedSGD(ps, lr=1e-2, momentum=0.9, weight_decay=1e-4)}
        opt = make[first](list(m.parameters()))
        opt.zero_grad()
        grads(m, 0)
        opt.step()
        ckpt_model = {k: v.clone() for k, v in m.state_dict().items()}
        ckpt_opt = opt.state_dict()
        # resume in a fresh model with the other optimiser
        m2, _ = build(1, 7, 'fp32')
        m2.train()
        m2.load_state_dict(ckpt_model)
        opt2 = make[second](list(m2.parameters()))
        opt2.load_state_dict(ckpt_opt)
        opt2.zero_grad()
        grads(m2, 1)
        opt2.step()
        return {k: v.detach().clone() for k, v in m2.named_parameters()}

    ref = run('torch', 'torch')
    for a, b in (('torch', 'fused'), ('fused', 'torch'), ('fused', 'fused')):
        got = run(a, b)
        for k in ref:
            scale = float(ref[k].abs().max())
            assert float((got[k] - ref[k]).abs().max()) <= 1e-5 * scale + 1e-9, (a, b, k)
    # the saved dictionary itself has torch's keys
    m, _ = build(1, 7, 'fp32')
    sd = FusedSGD(m.parameters()).state_dict()
    ref_keys = set(torch.optim.SGD(list(m.parameters()), lr=1e-3).state_dict()['param_groups'][0].keys())
    assert set(sd['param_groups'][0].keys()) == ref_keys and sd['state'] == {}


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_batch_of_two_clips_equals_two_single_clip_calls_in_eval(precision):
    """B = 2 (the reference always passes lists of length 1, but the API takes any B): in eval mode the clips of a
    batch do not interact, so each must come out exactly as if it had been passed alone -- both input layouts."""
    sp, fp = 3, 7
    m, dev = build(sp, fp, precision)
    m.eval()
    s0, f0 = clip_inputs(sp, fp, SMALL_LEVELS, 0, dev)
    s1, f1 = clip_inputs(sp, fp, SMALL_LEVELS, 1, dev)
    with torch.no_grad():
        both = m.temporally_enhance_features(s0 + s1, f0 + f1)
        one0 = m.temporally_enhance_features(s0, f0)
        one1 = m.temporally_enhance_features(s1, f1)
    for k in both:
        assert both[k].shape[0] == 2
        assert torch.equal(both[k][0:1], one0[k]), k
        assert torch.equal(both[k][1:2], one1[k]), k


@pytest.mark.parametrize('sp,fp', [(3, 7), (4, 32)])
def test_fp8_inference_path_error_is_measured_and_bounded(sp, fp):
    """precision='fp8' (BASELINE config 5, inference only): the four Cin = 256 convs (fast_conv1, slow_conv1-3 = 99 % of
    the forward FLOPs) on e4m3 operands -- input clip quantised per tensor (scale 32), weights per output channel, the
    slow pathway's concat buffers written as e4m3 by BN-apply (scale 32), f32 accumulate, bf16 results.  The tolerance
    is RE-STATED FROM MEASUREMENT, not assumed: rel-L2 and argmax agreement of the fused maps against the fp32 oracle
    are printed; the gates sit above the measurement: rel-L2 < 0.08, argmax agreement >= 0.90.  No input element may
    saturate at these scales.  Training / autograd state is refused."""
    m8, dev = build(sp, fp, 'fp8')
    mb, _ = build(sp, fp, 'bf16')
    m8.eval(); mb.eval()
    oracle = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    oracle.load_state_dict(closed_form_state_dict(oracle))
    oracle.eval()
    slow, fast = clip_inputs(sp, fp, SMALL_LEVELS, 0, dev)
    with torch.no_grad():
        o8 = m8.temporally_enhance_features(slow, fast)
        ob = mb.temporally_enhance_features(slow, fast)
        ref = oracle.temporally_enhance_features([OrderedDict((k, v.cpu()) for k, v in slow[0].items())],
                                                 [OrderedDict((k, v.cpu()) for k, v in fast[0].items())])
    for k in ref:
        a8, ab, r = o8[k].cpu(), ob[k].cpu(), ref[k]
        l2_8 = float((a8 - r).norm() / r.norm())
        l2_b = float((ab - r).norm() / r.norm())
        fast_8 = float((a8[:, 224:] - r[:, 224:]).norm() / r[:, 224:].norm())
        agree = float((a8.argmax(1) == r.argmax(1)).float().mean())
        print('(%d,%d) level %s: fp8 rel-L2 %.4f (fast channels %.4f), bf16 rel-L2 %.4f, argmax agreement fp8 %.3f'
              % (sp, fp, k, l2_8, fast_8, l2_b, agree))
        assert fast_8 < 0.08 and l2_8 < 0.08 and agree >= 0.90
    assert m8.fp8_saturated() == 0
    m8.train()
    with pytest.raises(RuntimeError):
        m8.temporally_enhance_features(slow, fast)


@pytest.mark.parametrize('fused,precision', [(True, 'fp32'), (False, 'fp32'), (True, 'bf16')])
def test_c1_eight_centre_frames_trajectory_matches_the_oracle(fused, precision):
    """BASELINE config 1 as SURVEY.md 8d restates it: the reference's default (sp, fp) = (1, 1) (constants.py:7-8), 8
    consecutive centre frames = 8 sequential B = 1 calls of temporally_enhance_features, backward after each, SGD
    (lr 1e-3, momentum 0.9, wd 1e-4) after every 2nd (model.py:318-323,369-374; train.py:80): FOUR optimiser steps of
    the module (fused optimiser with gradient sink, and torch.optim.SGD on the module's parameters) against the same
    trajectory of the CPU oracle -- every clip's loss, and after the 8th clip the accumulated update p - p_init of every
    parameter, every momentum buffer and every BN buffer (num_batches_tracked = 8 clips x 5 levels).
    Bounds (fp32): losses 1e-5; updates and momentum buffers 1e-2 rel-L2 -- measured 2e-3: gradients of clips without a
    ReLU-mask flip agree to 3e-5, a clip with a flipped element (a pre-activation within fp32 rounding of zero) moves
    the gradients behind it by up to 4e-3 of their scale; BN running statistics 1e-5.
    (Five levels whose smallest has 24 positions: BatchNorm over the 2 positions of a 1x2 level is ill-conditioned in
    BOTH implementations -- its backward is pure cancellation.)
    bf16 (the bench dtype; VERDICT r2 weak 2: "no statement of how a bf16 run drifts over optimiser steps"): the same
    trajectory with bf16 activations / fp32 master weights: the drift is MEASURED and printed -- per-clip loss error,
    accumulated update, momentum buffers, running statistics -- and bounded at 2e-2 (loss), 0.25 rel-L2 (updates,
    momentum), 2e-2 (running statistics)."""
    from sfvos_amd import FusedSGD, SlowFastLayers
    bf16 = precision == 'bf16'
    tol_loss, tol_u, tol_stat = (2e-2, 0.25, 2e-2) if bf16 else (1e-5, 1e-2, 1e-5)
    loss_errs = []
    dev = torch.device('cuda:0')
    shapes = OrderedDict([('0', (24, 42)), ('1', (12, 21)), ('2', (6, 11)), ('3', (5, 8)), ('pool', (4, 6))])
    g = torch.Generator().manual_seed(63)
    frames = OrderedDict((k, torch.randn(8, 256, h, w, generator=g)) for k, (h, w) in shapes.items())
    torch.manual_seed(5)
    ref = OracleSlowFastLayers(256, torch.device('cpu'), 1, 1)
    m = SlowFastLayers(256, dev, 1, 1, precision=precision)
    m.load_state_dict(ref.state_dict())
    m = m.to(dev)
    ref.train(); m.train()
    p0 = {n: q.detach().clone() for n, q in ref.named_parameters()}
    kw = dict(lr=1e-3, momentum=0.9, weight_decay=1e-4)
    opt_r = torch.optim.SGD(ref.parameters(), **kw)
    opt = FusedSGD(m.parameters(), **kw).attach(m) if fused else torch.optim.SGD(m.parameters(), **kw)
    opt.zero_grad(); opt_r.zero_grad()
    for i in range(8):
        win_r = OrderedDict((k, v[i:i + 1]) for k, v in frames.items())           # sp = fp = 1: the centre frame
        win = OrderedDict((k, v[i:i + 1].to(dev)) for k, v in frames.items())
        loss_r = proxy_loss(ref.temporally_enhance_features([win_r], [win_r]))
        loss_r.backward()
        loss = proxy_loss(m.temporally_enhance_features([win], [win]))
        loss.backward()
        lv, lr = float(loss.detach()), float(loss_r.detach())
        loss_errs.append(abs(lv - lr) / abs(lr))
        assert abs(lv - lr) <= tol_loss * abs(lr), (i, lv, lr)
        if i % 2 == 1:                                                           # model.py:372-374
            opt.step(); opt_r.step()
            opt.zero_grad(); opt_r.zero_grad()
    torch.cuda.synchronize()

    def rel_l2(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return float((a - b).norm() / b.norm().clamp_min(1e-30))

    zero_grad_bias = ('conv1.bias', 'conv2.bias', 'conv3.bias')   # conv bias in front of a train-mode BN: gradient 0
    worst_u = worst_m = worst_s = 0.0
    ref_p = dict(ref.named_parameters())
    sd_r, sd = opt_r.state_dict()['state'], opt.state_dict()['state']
    for i, (name, p) in enumerate(m.named_parameters()):
        if name.endswith(zero_grad_bias):
            # the update is weight decay on round-off: the parameter itself stays within 1e-6 of its scale
            a, b = p.detach().cpu().double(), ref_p[name].detach().double()
            assert float((a - b).abs().max()) <= (1e-4 if bf16 else 1e-6) * float(b.abs().max()) + 1e-9, name
            continue
        eu = rel_l2(p.detach().cpu() - p0[name], ref_p[name].detach() - p0[name])
        em = rel_l2(sd[i]['momentum_buffer'].cpu(), sd_r[i]['momentum_buffer'])
        worst_u, worst_m = max(worst_u, eu), max(worst_m, em)
        assert eu < tol_u and em < tol_u, (name, eu, em)
    ref_b = dict(ref.named_buffers())
    for name, bufr in m.named_buffers():
        if name.endswith('num_batches_tracked'):
            assert int(bufr) == int(ref_b[name]) == 8 * len(shapes)
        else:
            worst_s = max(worst_s, max_rel_err(bufr.cpu().numpy(), ref_b[name].numpy()))
            assert max_rel_err(bufr.cpu().numpy(), ref_b[name].numpy()) < tol_stat, name
    print('C1 trajectory (%s, %s): 8 clips, 4 optimiser steps: per-clip loss error %s; worst update rel-L2 %.2e, worst '
          'momentum buffer %.2e, worst running statistic %.2e'
          % ('FusedSGD + sink' if fused else 'torch.optim.SGD', precision, ' '.join('%.1e' % e for e in loss_errs),
             worst_u, worst_m, worst_s))
