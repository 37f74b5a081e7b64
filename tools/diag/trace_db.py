"""Per-kernel durations of a rocprofv3 --kernel-trace run stored as a rocpd sqlite database (ROCm 7.2 default output):
python tools/diag/trace_db.py <results.db> [substring]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ''
rows = list(db.cursor().execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count from kernels order by start"))
agg = collections.OrderedDict()
for n, d, g, w, l, v in rows:
    short = n.split('(')[0].replace('void ', '').replace('sfvos::', '')
    if sub not in short or short.startswith('at::') or 'at::native' in short:
        continue
    agg.setdefault((short[-70:], g // max(w, 1), l, v), []).append(d / 1000.0)
for (n, g, l, v), ds in agg.items():
    s = sorted(ds)
    print('%-72s wgs %6d lds %6d vgpr %3d  n=%3d  median %8.1f us  min %8.1f' % (n, g, l, v, len(s), s[len(s) // 2], s[0]))
