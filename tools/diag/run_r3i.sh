#!/bin/bash
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
OUT=gpurun_out/lwg_bench2.log; : > $OUT
for G in 1 2 4; do for W in 256 192; do
  SFVOS_LIB=$C/libsfvos_diag.so SFVOS_LWG_GROUPS=$G SFVOS_LWG_WGS=$W python bench.py --no-cpu-baseline --no-dropin --steps 6 --warmup 2 > gpurun_out/b_g.json 2>/dev/null
  python - <<PY >> $OUT
import json
d=json.loads(open('gpurun_out/b_g.json').read().strip().splitlines()[-1])
print('G $G W $W', {k:v['ms'] for k,v in d['hbm_layers'].items() if 'wgrad/l' in k})
PY
done; done
cat $OUT
