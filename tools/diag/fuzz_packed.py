"""Randomised PackedClip hand-overs (layouts, zero frames by pointer in front / behind, slow window offsets, B = 1-2)
against the same module called through temporally_enhance_features on explicitly zero-padded frame lists (fp32 / bf16;
forward outputs and every parameter gradient).  usage: python tools/diag/fuzz_packed.py [cases] [seed]"""
import os, random, sys
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sfvos_amd import PackedClip, SlowFastLayers

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda:0')
worst = 0.0
for case in range(n_cases):
    precision = rng.choice(['fp32', 'bf16'])
    fp = rng.randint(2, 12); sp = rng.randint(1, fp)
    B = rng.choice([1, 2])
    keys = ['0', '1', 'pool'][:rng.choice([1, 2, 3])]
    shapes = [(rng.randint(3, 30), rng.randint(3, 50)) for _ in keys]
    pad0 = rng.randint(0, max(0, fp - 2)); pad1 = rng.randint(0, max(0, fp - 1 - pad0 - 1))
    frames = fp - pad0 - pad1
    layout = 'grouped' if (precision == 'bf16' and rng.random() < 0.5) else 'ndhwc'
    slow_offset = rng.randint(0, fp - sp)
    torch.manual_seed(case)
    m = SlowFastLayers(256, dev, sp, fp, precision=precision).to(dev)
    m.train()
    tdt = torch.bfloat16 if precision == 'bf16' else torch.float32
    g = torch.Generator(device=dev).manual_seed(case)
    levels = [torch.randn((B, frames, h, w, 256), generator=g, device=dev).to(tdt) for (h, w) in shapes]
    clip = PackedClip.from_levels(levels, keys=keys, layout=layout, pad=(pad0, pad1))
    out = m.enhance_packed(clip, slow_offset)
    loss = sum((v.float() ** 2).mean() * (i + 1) for i, v in enumerate(out.values())) + sum(v.float().mean() for v in out.values())
    loss.backward()
    gp = {n: p.grad.clone() for n, p in m.named_parameters()}
    bufs = {n: b.clone() for n, b in m.named_buffers()}
    # the same through frame lists with materialised zero frames; fresh module with the same initial state
    torch.manual_seed(case)
    m2 = SlowFastLayers(256, dev, sp, fp, precision=precision).to(dev)
    m2.train()
    fast, slow = [], []
    for b in range(B):
        f = OrderedDict()
        for k, lv in zip(keys, levels):
            x = lv[b].permute(0, 3, 1, 2).float()                      # [frames, C, H, W]
            z0 = torch.zeros((pad0,) + tuple(x.shape[1:]), device=dev); z1 = torch.zeros((pad1,) + tuple(x.shape[1:]), device=dev)
            f[k] = torch.cat([z0, x, z1], 0)
        fast.append(f)
        slow.append(OrderedDict((k, v[slow_offset:slow_offset + sp]) for k, v in f.items()))
    out2 = m2.temporally_enhance_features(slow, fast)
    loss2 = sum((v.float() ** 2).mean() * (i + 1) for i, v in enumerate(out2.values())) + sum(v.float().mean() for v in out2.values())
    loss2.backward()
    eo = max(float((out[k] - out2[k]).abs().max() / out2[k].abs().max()) for k in keys)
    eg = 0.0
    for n, p in m2.named_parameters():
        d = float((gp[n] - p.grad).norm() / p.grad.norm().clamp_min(1e-20))
        if not n.endswith(('conv1.bias', 'conv2.bias', 'conv3.bias')):
            eg = max(eg, d)
    eb = max(float((bufs[n].float() - b.float()).abs().max()) for n, b in m2.named_buffers())
    worst = max(worst, eo, eg)
    flag = '' if (eo < 1e-5 and eg < 1e-4) else '   <-- DIFF'
    print('%-4s (%d,%d) B %d levels %-26s stored %2d pad (%d,%d) %-7s slow@%d: out %.1e grad %.1e buffers %.1e%s'
          % (precision, sp, fp, B, shapes, frames, pad0, pad1, layout, slow_offset, eo, eg, eb, flag), flush=True)
    if flag:
        sys.exit(1)
print('worst %.2e over %d cases' % (worst, n_cases))
