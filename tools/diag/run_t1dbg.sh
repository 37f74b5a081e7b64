#!/bin/bash
# wgrad_t1 kernel, timing-only switches (diagnostic library): what the 46 us are made of
cd $GRAFT_REPO_ROOT
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for d in 0 1 2 4 6 7 8 15; do
  echo "== debug $d"; SFVOS_T1_DEBUG=$d MB_COLD=1 timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 2>&1 | grep "f3"
done
