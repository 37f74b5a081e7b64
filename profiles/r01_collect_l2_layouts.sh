#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/l2; mkdir -p gpurun_out/l2
for mode in a b; do
  flag=""; [ $mode = a ] && flag="--ndhwc-input"
  timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/l2/${mode}1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $flag > gpurun_out/l2/${mode}1.log 2>&1 || exit 1
  timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/l2/${mode}2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $flag > gpurun_out/l2/${mode}2.log 2>&1 || exit 1
done
python - <<'PY'
import csv, glob, collections
for d in 'ab':
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/l2/%s[12]/*/*counter_collection.csv' % d):
        for r in csv.DictReader(open(f)):
            a[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    print('ndhwc' if d == 'a' else 'grouped')
    for k, cs in a.items():
        if 'conv3d_fs_kernel<1, 256>' in k or 'wgrad_kernel<1, 9, 1, 2, 4' in k or 'conv3d_kernel<1, 9, 3, 3' in k or 'wgrad_kernel<1, 9, 2, 2, 2' in k:
            print('  ', k.replace('sfvos::','')[:50], {c: '%.4g' % (sum(v)/len(v)) for c, v in cs.items()})
PY
