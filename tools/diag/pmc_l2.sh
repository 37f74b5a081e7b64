#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/l2; mkdir -p gpurun_out/l2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/l2/a -- python tools/diag/mb_conv.py ${1:-f1} 3 > gpurun_out/l2/a.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/l2/b -- python tools/diag/mb_conv.py ${1:-f1} 3 > gpurun_out/l2/b.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/l2/c -- python tools/diag/mb_conv.py ${1:-f1} 3 > gpurun_out/l2/c.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in 'abc':
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/l2/%s/*/*counter_collection.csv' % d):
        for r in csv.DictReader(open(f)):
            a[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in a.items():
        if 'sfvos' not in k: continue
        print(k.replace('sfvos::','')[:60], {c: '%.4g' % (sum(v)/len(v)) for c, v in cs.items()})
PY
tail -3 gpurun_out/l2/c.log
