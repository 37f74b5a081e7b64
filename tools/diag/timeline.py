import ctypes, os, sys, torch
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
NWG = 4096
buf = torch.zeros(8 * 128 * 4 + 64 + 4 * NWG, dtype=torch.int64, device='cuda')
os.environ['SFVOS_STAMP_PTR'] = hex(buf.data_ptr())
sys.argv = ['mb', sys.argv[1] if len(sys.argv) > 1 else 'f1', '3']
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mb_conv.py')).read())
torch.cuda.synchronize()
tl = buf.cpu().numpy()[8 * 128 * 4 + 64:].reshape(NWG, 4)
live = tl[:, 0] > 0
tl = tl[live]
t0 = tl[:, 0].min()
start = (tl[:, 0] - t0) / 100.0   # us
end = (tl[:, 1] - t0) / 100.0
dur = end - start
print('workgroups', len(tl), 'kernel span %.1f us' % end.max())
real = dur > 5.0
print('real workgroups', real.sum(), 'duration us: median %.1f mean %.1f p10 %.1f p90 %.1f max %.1f' % (
    np.median(dur[real]), dur[real].mean(), np.percentile(dur[real], 10), np.percentile(dur[real], 90), dur[real].max()))
hw = tl[:, 2]; xcc = tl[:, 3] & 0xf
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
cuid = xcc * 1000 + se * 100 + sh * 16 + cu
print('distinct CUs', len(set(cuid[real].tolist())))
# busy time per CU
busy = {}
for c, d in zip(cuid[real], dur[real]):
    busy[c] = busy.get(c, 0) + d
b = np.array(list(busy.values()))
print('per-CU busy us: min %.0f median %.0f max %.0f ; span %.0f -> mean utilisation %.3f' % (b.min(), np.median(b), b.max(), end.max(), b.sum() / (len(b) * end.max())))
# histogram of start times (rounds)
order = np.argsort(start)
for q in (0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0):
    i = order[min(len(order) - 1, int(q * (len(order) - 1)))]
    print('  q%.2f start %.1f us dur %.1f' % (q, start[i], dur[i]))
# duration by start-time bucket
edges = np.linspace(0, end.max(), 11)
for a_, b_ in zip(edges[:-1], edges[1:]):
    m = real & (start >= a_) & (start < b_)
    if m.sum():
        print('  started in [%.0f,%.0f) us: %d WGs, mean dur %.1f' % (a_, b_, m.sum(), dur[m].mean()))
