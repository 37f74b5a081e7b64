#!/bin/bash
# BN backward pass 2 in descending order (memory-side cache re-use of what pass 1 read last): tests, then A/B in the step
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "batchnorm or bn" 2>&1 | tail -2 || exit 1
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2 3; do
  for r in 0 1; do
    SFVOS_BN_REV=$r timeout -k 10 300 python bench.py --no-cpu-baseline --no-dropin --steps 20 --warmup 4 > gpurun_out/b_x.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('gpurun_out/b_x.json').read().strip().splitlines()[-1])
print('rev $r', d['value'], d['ms_per_step'], d['hbm_bound_passes']['bn_bwd'], {k:v[1] for k,v in d['kernels_ms'].items() if 'bn_bwd' in k})
PY
  done
done
