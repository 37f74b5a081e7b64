import ctypes, os, sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
buf = torch.zeros(8 * 128 * 4 + 64, dtype=torch.int64, device='cuda')
os.environ['SFVOS_STAMP_PTR'] = hex(buf.data_ptr())
sys.argv = ['mb', 'f1', '20']
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mb_conv.py')).read())
torch.cuda.synchronize()
import numpy as np
allb = buf.cpu().numpy()
st = allb[:8 * 128 * 4].reshape(8, 128, 4)
ck = allb[8 * 128 * 4:8 * 128 * 4 + 32].reshape(8, 4)
for wv in (0, 3, 4, 7):
    rows = st[wv]
    n = int((rows[:, 0] > 0).sum())
    P0, P1, P2, P3 = rows[:n, 0], rows[:n, 1], rows[:n, 2], rows[:n, 3]
    prep = (P1 - P0)[1:n]
    comp = (P2 - P1)[1:n]
    dmaw = (P3[1:n] - P2[:n-1])
    barw = (P0[1:n] - P3[1:n])
    tot = (P0[1:n] - P0[:n-1])
    dm, dr = ck[wv, 2] - ck[wv, 0], ck[wv, 3] - ck[wv, 1]
    print('wave', wv, 'stages', n, 'loop memtime ticks %d realtime ticks %d -> memtime/realtime %.3f' % (dm, dr, dm / max(dr, 1)))
    for name, v in (('prep', prep), ('compute', comp), ('vmcnt wait', dmaw), ('barrier wait', barw), ('stage total', tot)):
        print('  %-14s median %7.0f mean %7.0f  p90 %7.0f  max %7.0f (memtime ticks)' % (name, np.median(v), v.mean(), np.percentile(v, 90), v.max()))
    print('  first 14 stage totals', tot[:14].tolist())
