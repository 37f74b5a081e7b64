#!/bin/bash
cd $GRAFT_REPO_ROOT
python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/t5.log 2>&1; echo "exit $?" >> gpurun_out/t5.log
tail -4 gpurun_out/t5.log
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
OUT=gpurun_out/lwg_sweep4.log; : > $OUT
for G in 1 2 4; do
  for SUB in 1 2; do
    echo "== G $G SUB $SUB" >> $OUT
    MB_COLD=1 SFVOS_LWG_GROUPS=$G SFVOS_LWG_SUB=$SUB timeout -k 10 120 python tools/diag/mb_conv.py wlat 15 >> $OUT 2>&1 || exit 1
  done
done
echo "== default" >> $OUT
MB_COLD=1 timeout -k 10 120 python tools/diag/mb_conv.py wlat 15 >> $OUT 2>&1
# tall tiles A/B on the frame-split kernel (cold and warm)
for T in 0 1; do
  echo "== NO_TALL $T" >> $OUT
  if [ $T = 1 ]; then export SFVOS_NO_TALL_TILES=1; else unset SFVOS_NO_TALL_TILES; fi
  for i in 1 2; do
    timeout -k 10 120 python tools/diag/mb_conv.py f1 10 >> $OUT 2>&1
    timeout -k 10 120 python tools/diag/mb_conv.py f2 10 >> $OUT 2>&1
    timeout -k 10 120 python tools/diag/mb_conv.py df2 10 >> $OUT 2>&1
  done
done
grep -v amdgpu.ids $OUT | tail -60
