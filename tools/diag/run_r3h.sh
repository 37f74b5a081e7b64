#!/bin/bash
cd $GRAFT_REPO_ROOT
python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "lateral or wgrad" > gpurun_out/t6.log 2>&1; echo "exit $?" >> gpurun_out/t6.log
tail -3 gpurun_out/t6.log
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
OUT=gpurun_out/lwg_dbg2.log; : > $OUT
for D in 0 1 2 3; do
  echo "== DEBUG $D" >> $OUT
  SFVOS_LWG_DEBUG=$D timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 2>&1 | grep "l1\|l2" >> $OUT
done
for S in 1 2; do for R in 2 3 4; do
  echo "== SUB $S RING $R" >> $OUT
  SFVOS_LWG_SUB=$S SFVOS_LWG_RING=$R timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 2>&1 | grep "l1\|l2" >> $OUT
done; done
unset SFVOS_LIB
python bench.py --no-cpu-baseline --no-dropin --steps 6 --warmup 2 > gpurun_out/b_x.json 2>/dev/null
python - <<PY >> $OUT
import json
d=json.loads(open('gpurun_out/b_x.json').read().strip().splitlines()[-1])
print('bench', {k:(v['ms'],v['frac']) for k,v in d['hbm_layers'].items() if 'wgrad' in k})
PY
cat $OUT
