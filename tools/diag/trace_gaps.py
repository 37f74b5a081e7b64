"""Idle time of the GPU inside the timed steps, from a rocprofv3 --kernel-trace CSV: union of the kernel intervals vs
wall time between the first and last kernel of the analysed window.
usage: python tools/diag/trace_gaps.py <kernel_trace.csv> [first_f1_index last_f1_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda t: t[0])
f1 = [i for i, k in enumerate(ks) if 'conv3d_fs_kernel<1, 256>' in k[2]]
a = int(sys.argv[2]) if len(sys.argv) > 2 else len(f1) // 2
b = int(sys.argv[3]) if len(sys.argv) > 3 else a + 4
win = ks[f1[a]:f1[b]]
t0, t1 = win[0][0], max(k[1] for k in win)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
steps = b - a
print('steps %d: wall %.3f ms/step, GPU busy (union of kernels) %.3f ms/step, idle %.3f ms/step, sum of kernel durations %.3f ms/step'
      % (steps, (t1 - t0) / steps / 1e6, busy / steps / 1e6, (t1 - t0 - busy) / steps / 1e6,
         sum(e - s for s, e, _ in win) / steps / 1e6))
# largest idle gaps
gaps = []
cur_e = win[0][1]
for s, e, n in win[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, n))
    cur_e = max(cur_e, e)
gaps.sort(reverse=True)
for g, n in gaps[:12]:
    print('  gap %7.1f us before %s' % (g / 1e3, n[:90]))
print('  gaps > 1 us: %d, total %.3f ms/step' % (sum(1 for g, _ in gaps if g > 1000), sum(g for g, _ in gaps) / steps / 1e6))
