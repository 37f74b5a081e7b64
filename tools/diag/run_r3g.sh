#!/bin/bash
cd $GRAFT_REPO_ROOT
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
OUT=gpurun_out/lwg_dbg.log; : > $OUT
for G in 1 2; do
for D in 0 1 2 3 4 5 7; do
  echo "== G $G DEBUG $D" >> $OUT
  SFVOS_LWG_GROUPS=$G SFVOS_LWG_DEBUG=$D timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 2>&1 | grep "l1\|l2" >> $OUT
done
done
cat $OUT
