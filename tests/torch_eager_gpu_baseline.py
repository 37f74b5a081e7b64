"""Information only (not a test, not collected by pytest): how long the same fwd+bwd+SGD step takes in plain
PyTorch-ROCm eager on this GPU -- the CPU oracle module moved to cuda:0 (MIOpen conv3d, native batch-norm).
    MIOPEN_FIND_MODE=FAST python tests/torch_eager_gpu_baseline.py
Log of one run: profiles/r01_torch_eager_gpu_baseline.log"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.slowfast_ref import OracleSlowFastLayers, proxy_loss
from sfvos_amd import davis_pyramid
dev = torch.device('cuda:0')
sp, fp = 4, 32
pyr = davis_pyramid()
for dtype in (torch.bfloat16, torch.float32):
    torch.manual_seed(0)
    m = OracleSlowFastLayers(256, dev, sp, fp).to(dev).train()
    if dtype == torch.bfloat16:
        m = m.to(torch.bfloat16)
    fast = [torch.randn(1, 256, fp, h, w, device=dev, dtype=dtype) for _, (h, w) in pyr]
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    def step():
        total = 0
        for li, f in enumerate(fast):
            s = f[:, :, fp // 2 - sp // 2: fp // 2 + (sp + 1) // 2]
            so, fo = m(s, f)
            total = total + (torch.cat([so, fo], 1).float() ** 2).mean()
        total.backward()
        opt.step(); opt.zero_grad()
    try:
        for i in range(2):
            step(); torch.cuda.synchronize(); print('warm-up step', i, 'done', flush=True)
        t0 = time.time()
        n = 5
        for _ in range(n): step()
        torch.cuda.synchronize()
        print(dtype, 'torch eager (MIOpen) ms/step %.1f' % ((time.time() - t0) / n * 1e3), flush=True)
    except Exception as e:
        print(dtype, 'failed:', repr(e)[:300], flush=True)
    del m, fast, opt
    torch.cuda.empty_cache()
