import json
import sys

# usage: show_bench.py [bench-line.json] [rows]   (no file: the line is read from stdin)
args = sys.argv[1:]
src = open(args.pop(0)).read() if args and not args[0].isdigit() else sys.stdin.read()
d = json.loads(src.strip().splitlines()[-1])
print('clips/s %.2f  ms/step %.2f  whole-step TF/s %.1f' % (d['value'], d['ms_per_step'], d['achieved_tflops_whole_step']))
print('roofline', d['roofline'])
k = d['kernels_ms']
groups = {}
steps = (d.get('kernel_events') or {}).get('layer_table_steps') or d['steps']
for n, (c, t) in k.items():
    g = n.split('/')[0]
    groups[g] = groups.get(g, 0.0) + c * t / steps
print('per-step ms by kind (top-24 only):', {g: round(v, 2) for g, v in groups.items()})
for n, (c, t) in list(k.items())[:int(args[0]) if args else 14]:
    print('%-28s %8.3f' % (n, t))
