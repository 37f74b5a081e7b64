#!/bin/bash
# fast_conv2 weight gradient: 4 taps x 2 row halves per workgroup (8- or 16-row tiles) against the 8-tap configuration
cd $GRAFT_REPO_ROOT
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2 3; do
  echo "== 4 taps x 2 row halves, TH 8"; timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep " f2 "
  echo "== 4 taps x 2 row halves, TH 16"; SFVOS_WGRAD_TH16=1 timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep " f2 "
  echo "== 8 taps"; SFVOS_WGRAD_KS1=1 timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep " f2 "
done
