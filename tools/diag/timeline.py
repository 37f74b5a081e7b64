"""Timeline of one steady-state bench step from a rocprofv3 --kernel-trace CSV (two HIP streams): per stream the busy
time and the gaps, the union busy time, and the kernels in start order with (stream, start, duration, overlap partner).
usage: python tools/diag/timeline.py <kernel_trace.csv> [step_index]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r.get('Queue_Id', r.get('Stream_Id', 0)) or 0),
                 n.split('(')[0].replace('void ', '').replace('sfvos::', '')[:48]))
rows.sort()
# steps are delimited by the dominant kernel (fast_conv1 forward: conv3d_fs_kernel<1, 256>)
starts = [i for i, r in enumerate(rows) if 'conv3d_fs_kernel<1, 256>' in r[3]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 3
a, b = starts[k], starts[k + 1]
# the step really begins a few small launches earlier (layout, pack); good enough: dominant kernel to dominant kernel
seg = rows[a:b]
t0 = seg[0][0]
T = rows[b][0] - t0
print('step %d: %.3f ms between two fast_conv1 forward launches, %d kernels' % (k, T / 1e6, len(seg)))
ev = []
for s, e, q, n in seg:
    ev.append((s, 1)); ev.append((min(e, rows[b][0]), -1))
ev.sort()
busy = 0; depth = 0; last = t0; two = 0
for t, d in ev:
    if depth > 0: busy += t - last
    if depth > 1: two += t - last
    depth += d; last = t
print('union busy %.3f ms (%.1f %%), two or more kernels in flight %.3f ms' % (busy / 1e6, 100.0 * busy / T, two / 1e6))
perq = collections.defaultdict(int)
for s, e, q, n in seg: perq[q] += e - s
print('busy per queue (ms):', {q: round(v / 1e6, 3) for q, v in perq.items()})
print('sum of durations %.3f ms' % (sum(e - s for s, e, q, n in seg) / 1e6))
for s, e, q, n in seg:
    if e - s > 15000:
        print('%9.1f us  q%-2d %8.1f us  %s' % ((s - t0) / 1e3, q, (e - s) / 1e3, n))
