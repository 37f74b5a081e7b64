#!/bin/bash
cd $GRAFT_REPO_ROOT
python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/t7.log 2>&1; echo "exit $?" >> gpurun_out/t7.log
tail -3 gpurun_out/t7.log
timeout -k 10 120 python tools/diag/mb_conv.py f3 20 2>&1 | grep conv
python bench.py --no-cpu-baseline --no-dropin --steps 10 --warmup 2 > gpurun_out/b_x.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/b_x.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print('bench', {k:(v['ms'],v['frac']) for k,v in d['hbm_layers'].items()})
PY
