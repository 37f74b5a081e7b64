"""GPU: sequence-mode inference with sliding-window reuse (sfvos_amd.SlowFastStream) against the CPU
ORACLE (oracle/slowfast_ref.py, eval mode) run on every window of a short video, i.e. against what the reference's
per-frame loop computes (code/helpers/model.py:316-340: window of fp frames around each frame, zero frames beyond
the ends, model.py:215-225).  fp32: 1e-3 of the tensor scale and identical argmax (north_star tolerance); bf16:
error measured against the fp32 oracle, bounded at 5e-2.  Extras: against the module's own per-window
temporally_enhance_features the stream is within 1e-5 (fp32) / bit-identical (bf16: same kernels, same operands,
same summation order)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from golden_util import SMALL_LEVELS
from oracle.closed_form import closed_form_features, closed_form_state_dict
from oracle.slowfast_ref import OracleSlowFastLayers

pytestmark = pytest.mark.gpu


def build(sp, fp, precision):
    from sfvos_amd import SlowFastLayers
    dev = torch.device('cuda:0')
    m = SlowFastLayers(256, dev, sp, fp, precision=precision)
    m.load_state_dict(closed_form_state_dict(m))
    return m.to(dev).eval(), dev


def reference_windows(m, frames, sp, fp):
    """What the reference loop feeds the module for each centre frame, and what it gets back: from `m` (the
    module under test on the GPU, or the CPU oracle)."""
    N = len(frames)
    outs = []
    zero = OrderedDict((k, torch.zeros_like(v)) for k, v in frames[0].items())
    with torch.no_grad():
        for i in range(N):
            idx = range(i - fp // 2, i - fp // 2 + fp)
            win = [frames[j] if 0 <= j < N else zero for j in idx]
            fast = OrderedDict((k, torch.stack([w[k] for w in win])) for k in frames[0])        # [fp,256,H,W]
            c = fp // 2
            slow = OrderedDict((k, v[c - sp // 2: c + (sp + 1) // 2]) for k, v in fast.items())
            outs.append(m.temporally_enhance_features([slow], [fast]))
    return outs


def oracle_windows(frames_cpu, sp, fp):
    o = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    o.load_state_dict(closed_form_state_dict(o))
    o.eval()
    return reference_windows(o, frames_cpu, sp, fp)


@pytest.mark.parametrize('sp,fp,precision,chunk', [(3, 7, 'fp32', 1), (1, 1, 'fp32', 2), (1, 7, 'bf16', 1),
                                                   (4, 32, 'bf16', 1), (7, 7, 'fp32', 3), (3, 7, 'bf16', 4),
                                                   (4, 32, 'bf16', 4), (3, 7, 'fp32', 5)])
def test_stream_equals_per_window_recompute(sp, fp, precision, chunk):
    from sfvos_amd import SlowFastStream
    m, dev = build(sp, fp, precision)
    N = 7 if fp < 32 else 5
    seq = closed_form_features(N, SMALL_LEVELS, clip=11)                      # level -> [N,256,H,W]
    frames = [OrderedDict((k, v[i].to(dev)) for k, v in seq.items()) for i in range(N)]
    oref = oracle_windows([OrderedDict((k, v[i]) for k, v in seq.items()) for i in range(N)], sp, fp)
    ref = reference_windows(m, frames, sp, fp)
    stream = SlowFastStream(m, list(SMALL_LEVELS.values()), keys=list(SMALL_LEVELS.keys()), chunk=chunk)
    got = stream.run_sequence(frames)
    assert len(got) == N
    worst = 0.0
    for i in range(N):
        for k in ref[i]:
            a, b, o = got[i][k], ref[i][k], oref[i][k]
            assert a.shape == b.shape == o.shape and a.dtype == torch.float32
            # against the CPU oracle (the parity gate)
            e = float((a.cpu() - o).abs().max() / o.abs().max())
            worst = max(worst, e)
            if precision == 'bf16':
                assert e < 5e-2, (i, k, e)
            else:
                assert e < 1e-3, (i, k, e)
                # argmax over the 256 fused channels: identical wherever the oracle's own top-2 margin exceeds fp32
                # round-off (1e-4 of the scale)
                top2 = o.topk(2, dim=1).values
                safe = (top2[:, 0] - top2[:, 1]) > 1e-4 * float(o.abs().max())
                assert bool((a.argmax(1).cpu() == o.argmax(1))[safe].all()), (i, k)
            # extras: against the module's own per-window recompute
            if precision == 'bf16':
                assert torch.equal(a, b), (i, k)
            else:
                scale = float(b.abs().max())
                assert float((a - b).abs().max()) <= 1e-5 * scale, (i, k)
                assert torch.equal(a.argmax(1), b.argmax(1))
    print('stream (%d,%d) %s chunk %d: worst max err / scale vs the CPU oracle %.3e' % (sp, fp, precision, chunk, worst))
    # a second sequence through the same object (reset) gives the same answers
    again = stream.run_sequence(frames)
    for i in range(N):
        for k in ref[i]:
            assert torch.equal(again[i][k], got[i][k])


def test_stream_refuses_train_mode_and_cpu():
    from sfvos_amd import SlowFastLayers, SlowFastStream
    m, dev = build(1, 1, 'fp32')
    s = SlowFastStream(m, list(SMALL_LEVELS.values()), keys=list(SMALL_LEVELS.keys()))
    m.train()
    with pytest.raises(RuntimeError):
        s.push(None)
    with pytest.raises(RuntimeError):
        SlowFastStream(SlowFastLayers(256, torch.device('cpu'), 1, 1), [(4, 4)])
