#!/bin/bash
# fast_conv1 weight gradient: (1 n-tile, 1 c-tile, 4 taps) x 2 row halves on 16-row tiles against (1, 2 c-tiles, 4 taps)
cd $GRAFT_REPO_ROOT
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2 3; do
  echo "== shipped"; timeout -k 10 120 python tools/diag/mb_conv.py wf1 10 2>&1 | grep " f1 "
  echo "== row split"; SFVOS_WGRAD_F1KS=1 timeout -k 10 120 python tools/diag/mb_conv.py wf1 10 2>&1 | grep " f1 "
done
