#!/bin/bash
# A/B of the BN-backward block reduction (16 KB table vs shuffles + 4 KB): tests, then the step time with each library.
cd $GRAFT_REPO_ROOT
python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "batchnorm or bn" 2>&1 | tail -2
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_bnold.so; else unset SFVOS_LIB; fi
    python bench.py --no-cpu-baseline --no-dropin --steps 20 --warmup 3 > gpurun_out/b_x.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('gpurun_out/b_x.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], {k:v[1] for k,v in d['kernels_ms'].items() if 'bn_bwd' in k})
PY
  done
done
