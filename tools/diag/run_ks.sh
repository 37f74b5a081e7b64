#!/bin/bash
# fast_conv2 weight gradient: 4 taps x 2 row halves per workgroup on 16-row tiles (shipped) against the same on 8-row tiles
# and against 8 taps per workgroup (switches of the diagnostic library)
cd $GRAFT_REPO_ROOT
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2 3; do
  echo "== 4 taps x 2 row halves, 16-row tiles"; timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep " f2 "
  echo "== 4 taps x 2 row halves, 8-row tiles"; SFVOS_WGRAD_TH8=1 timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep " f2 "
  echo "== 8 taps"; SFVOS_WGRAD_KS1=1 timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep " f2 "
done
