#!/bin/bash
# kernel-trace stats + PMC traffic passes of the SAME bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profb; mkdir -p gpurun_out/profb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profb/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/profb/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/profb/fetch -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/profb/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/profb/write -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/profb/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/profb/mfma -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/profb/mfma.log 2>&1
python - <<'PY'
import csv, glob, collections, json
def agg(d):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/profb/%s/*/*counter_collection.csv' % d):
        for r in csv.DictReader(open(f)):
            a[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return a
out = {}
for d in ('fetch', 'write', 'mfma'):
    for k, cs in agg(d).items():
        if 'sfvos' not in k: continue
        name = k.replace('sfvos::', '').split('(')[0].replace('void ', '')
        for c, v in cs.items():
            out.setdefault(name, {})[c] = {'launches': len(v), 'mean': sum(v) / len(v), 'max': max(v)}
json.dump(out, open('gpurun_out/profb/pmc_summary.json', 'w'), indent=1, sort_keys=True)
for name in sorted(out, key=lambda n: -out[n].get('FETCH_SIZE', {}).get('max', 0))[:8]:
    print(name[:60], {c: '%.4g' % v['max'] for c, v in out[name].items()})
PY
f=$(ls gpurun_out/profb/trace/*/*kernel_stats.csv | head -1); cp $f gpurun_out/profb/kernel_stats.csv; head -12 $f | cut -c1-160
