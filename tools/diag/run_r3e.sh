#!/bin/bash
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
OUT=gpurun_out/ab_tall.log; : > $OUT
for i in 1 2 3; do
  for L in libsfvos.so libsfvos_notall.so; do
    echo "== $L" >> $OUT
    SFVOS_LIB=$C/$L timeout -k 10 120 python tools/diag/mb_conv.py f1 10 >> $OUT 2>&1 || exit 1
    SFVOS_LIB=$C/$L timeout -k 10 120 python tools/diag/mb_conv.py f2 10 >> $OUT 2>&1
    SFVOS_LIB=$C/$L timeout -k 10 120 python tools/diag/mb_conv.py df2 10 >> $OUT 2>&1
    SFVOS_LIB=$C/$L timeout -k 10 120 python tools/diag/mb_conv.py f3 10 >> $OUT 2>&1
  done
done
grep -v amdgpu.ids $OUT
OUT=gpurun_out/lwg_bench.log; : > $OUT
for G in 1 2 4; do
  SFVOS_LIB=$C/libsfvos_diag.so SFVOS_LWG_GROUPS=$G python bench.py --no-cpu-baseline --no-dropin --steps 6 --warmup 2 > gpurun_out/b_g$G.json 2>> $OUT
  python - <<PY >> $OUT
import json
d=json.loads(open('gpurun_out/b_g$G.json').read().strip().splitlines()[-1])
print('G $G', {k:v['ms'] for k,v in d['hbm_layers'].items() if 'wgrad' in k})
PY
done
cat $OUT
