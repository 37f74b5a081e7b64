"""Experiment: the bench step captured into hipGraphs (torch.cuda.graph), 1 or 2 HIP streams per clip.
usage: python tools/diag/graph_step.py [streams] [steps] [sp] [fp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from sfvos_amd import FusedSGD, GradBucket, MSEProxyLoss, PackedClip, SlowFastLayers, davis_pyramid

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SP = int(sys.argv[3]) if len(sys.argv) > 3 else 4
FP = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device('cuda', 0)
torch.manual_seed(63)
model = SlowFastLayers(256, dev, SP, FP, precision='bf16').to(dev)
model.train()
model.n_streams = streams
opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
bucket = GradBucket(opt.flat_grad)
opt.attach(model, bucket)
pyr = davis_pyramid()
gen = torch.Generator(device=dev).manual_seed(63)
levels = [torch.randn((1, FP, h, w, 256), generator=gen, device=dev).bfloat16() for _, (h, w) in pyr]
clip = PackedClip.from_levels(levels, keys=[k for k, _ in pyr], layout='grouped')
del levels
loss_fn = MSEProxyLoss({k: torch.randn((1, 256, h, w), generator=gen, device=dev) for k, (h, w) in pyr})
step = bench.make_step(model, opt, bucket, loss_fn, lambda: model.enhance_packed(clip))

def timed(f, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        f(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

for i in range(6):
    step(i)
print('eager   streams=%d  %.3f ms/step' % (streams, timed(step, steps)), flush=True)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
graphs, losses = [], []
with torch.cuda.stream(s):
    for i in range(2):
        step(i)       # warm-up on the capture stream
    for parity in (0, 1):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            losses.append(step(parity))
        graphs.append(g)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print('captured', flush=True)
replay = lambda i: graphs[i % 2].replay()
for i in range(4):
    replay(i)
print('graph   streams=%d  %.3f ms/step   loss %.6f' % (streams, timed(replay, steps), float(losses[1])), flush=True)
