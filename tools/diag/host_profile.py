"""Host-side profile of the eager step (cProfile) on a launch-bound configuration.
usage: python tools/diag/host_profile.py [sp] [fp] [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from sfvos_amd import FusedSGD, GradBucket, MSEProxyLoss, PackedClip, SlowFastLayers, davis_pyramid

SP = int(sys.argv[1]) if len(sys.argv) > 1 else 1
FP = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device('cuda', 0)
torch.manual_seed(63)
model = SlowFastLayers(256, dev, SP, FP, precision='bf16').to(dev)
model.train()
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
for i in range(6):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('(%d,%d): host enqueue %.3f ms/step, with GPU drain %.3f ms/step' % (SP, FP, 1e3 * t_host / steps, 1e3 * t_all / steps))
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(35)
