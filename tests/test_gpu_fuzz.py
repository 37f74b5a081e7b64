"""GPU: randomised configurations of sfvos_amd.SlowFastLayers against the CPU oracle (oracle/slowfast_ref.py).

The fixtures and the full-size tests pin the configurations the reference's tables name; this test draws others: any
(slow, fast) pathway sizes with slow <= fast <= 12, pyramids of 1-3 levels with sizes that are not multiples of any tile, one
or two clips per call, default (random) initialisation, N(0,1) features -- train-mode forward, the proxy loss, every
parameter gradient, the BN buffers, and an eval forward.  Seeds 0 .. SFVOS_FUZZ-1 (default 4; the cases are cheap on the
GPU, the oracle is the cost: a few seconds each).  Tolerances: fp32 fused maps 1e-4 of the map's scale; layer-3 gradients
(nothing discontinuous behind them) 1e-4 rel-L2; layer-1/2 and lateral gradients 3e-2 -- a ReLU mask that flips between
two correct implementations (pre-activation within round-off of 0) moves single entries of everything in front of it, the
effect tests/test_gpu_parity.py measures at full size (same gate there); bf16 fused maps 3e-2, gradients 0.2."""
import os
import random
from collections import OrderedDict

import pytest
import torch

from oracle.slowfast_ref import OracleSlowFastLayers, proxy_loss

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get('SFVOS_FUZZ', '4'))
MAX_FP = int(os.environ.get('SFVOS_FUZZ_MAXFP', '12'))   # larger windows (temporal kernels up to 23, laterals up to 45): run by hand


def _rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('seed', range(N_CASES))
def test_random_configuration_matches_the_oracle(seed, precision):
    from sfvos_amd import SlowFastLayers
    rng = random.Random(1234 + seed)
    fp = rng.randint(1, MAX_FP)
    sp = rng.randint(1, fp)
    B = rng.choice([1, 1, 2])
    keys = ['0', '1', 'pool'][:rng.choice([1, 2, 3])]
    shapes = [(rng.randint(3, 40 if MAX_FP <= 12 else 14), rng.randint(3, 70 if MAX_FP <= 12 else 36)) for _ in keys]
    dev = torch.device('cuda:0')
    torch.manual_seed(seed)
    m = SlowFastLayers(256, dev, sp, fp, precision=precision).to(dev)
    o = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    o.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    g = torch.Generator().manual_seed(seed)
    fast = [OrderedDict((k, torch.randn(fp, 256, h, w, generator=g)) for k, (h, w) in zip(keys, shapes)) for _ in range(B)]
    lo = fp // 2 - sp // 2   # the centre frames, as SegmentationModel._slice_features takes them (model.py:242-248)
    slow = [OrderedDict((k, v[lo:lo + sp]) for k, v in f.items()) for f in fast]
    to_dev = lambda clips: [OrderedDict((k, v.to(dev)) for k, v in c.items()) for c in clips]

    m.train(); o.train()
    out = m.temporally_enhance_features(to_dev(slow), to_dev(fast))
    ref = o.temporally_enhance_features(slow, fast)
    loss, rloss = proxy_loss(out), proxy_loss(ref)
    loss.backward(); rloss.backward()
    tol_out, tol_grad, tol_grad3 = (1e-4, 3e-2, 1e-4) if precision == 'fp32' else (3e-2, 0.2, 0.2)
    worst_out = 0.0
    for k in keys:
        assert out[k].shape == ref[k].shape
        e = float((out[k].detach().cpu().double() - ref[k].detach().double()).abs().max() / ref[k].detach().abs().max())
        worst_out = max(worst_out, e)
        assert e < tol_out, 'level %s: %.2e' % (k, e)
    assert abs(float(loss.detach()) - float(rloss.detach())) < 10 * tol_out * abs(float(rloss.detach()))
    worst_grad = 0.0
    rp = dict(o.named_parameters())
    gscale = max(float(p.grad.norm()) for p in rp.values())
    for name, p in m.named_parameters():
        if name.endswith(('conv1.bias', 'conv2.bias', 'conv3.bias')):
            # a conv bias in front of a train-mode BN: the true gradient is 0, both sides hold round-off
            assert float(p.grad.norm()) < (1e-4 if precision == 'fp32' else 2e-2) * gscale
            continue
        e = _rel_l2(p.grad.detach().cpu(), rp[name].grad)
        worst_grad = max(worst_grad, e)
        layer3 = name.startswith(('slow_conv3', 'fast_conv3', 'bn_s3', 'bn_f3'))
        assert e < (tol_grad3 if layer3 else tol_grad), '%s: %.2e' % (name, e)
    rb = dict(o.named_buffers())
    for name, b in m.named_buffers():
        if name.endswith('num_batches_tracked'):
            assert int(b) == int(rb[name])
        else:
            assert _rel_l2(b.detach().cpu(), rb[name]) < 10 * tol_out, name
    m.eval(); o.eval()
    with torch.no_grad():
        eo = m.temporally_enhance_features(to_dev(slow), to_dev(fast))
        er = o.temporally_enhance_features(slow, fast)
    for k in keys:
        e = float((eo[k].cpu().double() - er[k].double()).abs().max() / er[k].abs().max())
        assert e < tol_out, 'eval level %s: %.2e' % (k, e)
    print('seed %d %s (sp,fp)=(%d,%d) B=%d levels %s: fused maps %.1e, worst gradient %.1e'
          % (seed, precision, sp, fp, B, shapes, worst_out, worst_grad))
