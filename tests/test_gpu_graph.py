"""GraphedStep (sfvos_amd/graph.py): the accumulate-2 training cycle of reference code/helpers/model.py:340-374 replayed
from hipGraphs must be the eager cycle bit for bit -- same kernels, same order, same addresses -- and constructing it must
not train.  (What the eager cycle computes is checked against the oracle in tests/test_gpu_parity.py.)"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(sp, fp, precision, seed=5):
    from sfvos_amd import FusedSGD, MSEProxyLoss, PackedClip, SlowFastLayers
    dev = torch.device('cuda', 0)
    torch.manual_seed(seed)
    model = SlowFastLayers(256, dev, sp, fp, precision=precision).to(dev)
    model.train()
    opt = FusedSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.attach(model)
    shapes = [('0', (24, 42)), ('1', (12, 21)), ('pool', (6, 11))]
    tdt = torch.bfloat16 if precision == 'bf16' else torch.float32
    gen = torch.Generator(device=dev).manual_seed(seed)
    clips = []
    for _ in range(6):
        levels = [torch.randn((1, fp, h, w, 256), generator=gen, device=dev).to(tdt) for _, (h, w) in shapes]
        clips.append(PackedClip.from_levels(levels, keys=[k for k, _ in shapes], layout='ndhwc'))
    loss_fn = MSEProxyLoss({k: torch.randn((1, 256, h, w), generator=gen, device=dev) for k, (h, w) in shapes})
    return model, opt, clips, loss_fn


def _state(model, opt):
    st = {k: v.detach().clone() for k, v in model.state_dict().items()}
    st['__momentum'] = opt.flat_buf.clone()
    return st


@pytest.mark.parametrize('cfg', [(1, 1, 'bf16'), (3, 7, 'bf16'), (3, 7, 'fp32')])
def test_graphed_cycle_is_the_eager_cycle_bit_for_bit(cfg):
    from sfvos_amd import GraphedStep, PackedClip
    sp, fp, precision = cfg
    # eager: six clips = three optimiser steps
    model, opt, clips, loss_fn = _setup(sp, fp, precision)
    eager_losses = []
    for i, clip in enumerate(clips):
        loss = loss_fn(model.enhance_packed(clip))
        loss.backward()
        eager_losses.append(float(loss.detach()))
        if i % 2 == 1:
            opt.step()
            opt.zero_grad()
    want = _state(model, opt)

    # graphs: the same six clips copied into the static buffer
    model, opt, clips, loss_fn = _setup(sp, fp, precision)
    before = _state(model, opt)
    static = PackedClip(clips[0].data.clone(), clips[0].shapes, clips[0].batch, clips[0].frames, clips[0].keys)
    step = GraphedStep(model, opt, loss_fn, static, accumulate=2)
    after = _state(model, opt)
    for k in before:
        assert torch.equal(before[k], after[k]), 'constructing a GraphedStep changed %s' % k
    losses = []
    for clip in clips:
        static.data.copy_(clip.data)
        losses.append(float(step()))
    got = _state(model, opt)
    print('%s losses eager %s graph %s' % (cfg, ['%.6f' % v for v in eager_losses], ['%.6f' % v for v in losses]))
    assert losses == eager_losses
    for k in want:
        assert torch.equal(want[k], got[k]), '%s differs between the eager and the replayed cycle' % k
    assert abs(eager_losses[-1] - eager_losses[0]) > 0, 'the parameters never moved'

    # the eager path after replays: its cached weight images are stale (the SGD steps ran inside the graphs) and must be
    # re-packed -- an eval forward right after equals one of a fresh module holding the same state
    from sfvos_amd import SlowFastLayers
    model.eval()
    with torch.no_grad():
        a = model.enhance_packed(clips[0])
    fresh = SlowFastLayers(256, torch.device('cuda', 0), sp, fp, precision=precision).to('cuda:0')
    fresh.load_state_dict(model.state_dict())
    fresh.eval()
    with torch.no_grad():
        b = fresh.enhance_packed(clips[0])
    for k in a:
        assert torch.equal(a[k], b[k]), 'stale packed weights after graph replays (level %s)' % k


def test_graphed_step_refuses_parameter_changes_in_mid_cycle():
    from sfvos_amd import GraphedStep
    model, opt, clips, loss_fn = _setup(1, 1, 'bf16')
    step = GraphedStep(model, opt, loss_fn, clips[0], accumulate=2)
    step()                       # clip 1 of the cycle
    opt.step()                   # an eager optimiser step behind the graphs' back
    with pytest.raises(RuntimeError, match='middle of an accumulation cycle'):
        step()
    step.reset()
    step()
    step()
    assert step.position == 0
