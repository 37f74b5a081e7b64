#!/bin/bash
# Every (slow, fast) configuration the reference's tables name, through bench.py: step time and the per-layer fractions.
cd $GRAFT_REPO_ROOT
for cfg in "1 1" "3 3" "7 7" "1 7" "3 7" "1 8" "8 8"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --sp $1 --fp $2 --no-cpu-baseline --no-dropin --steps 20 --warmup 3 > gpurun_out/b_cfg_$1_$2.json 2>gpurun_out/b_cfg_$1_$2.err || { echo "($1,$2) FAILED"; tail -3 gpurun_out/b_cfg_$1_$2.err; continue; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/b_cfg_$1_$2.json').read().strip().splitlines()[-1])
print('($1,$2)', d['value'], 'clips/s', d['ms_per_step'], 'ms', 'roofline', d['roofline']['frac'])
for key in ('mfma_layers','hbm_layers'):
    print('  ', key, {k:(round(v['frac'],3) if isinstance(v,dict) else v) for k,v in d.get(key,{}).items()})
PY
done
