#!/bin/bash
# Timing-only builds of conv3d.hip (-DSFVOS_ABLATE=1 no A re-reads, 2 no B re-reads, 3 neither; results wrong):
# how much of the wide kernel's time is the LDS operand reads.
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
for i in 1 2; do
  for v in "" abl1 abl2 abl3; do
    if [ -z "$v" ]; then unset SFVOS_LIB; else export SFVOS_LIB=$L/libsfvos_$v.so; fi
    echo "== lib ${v:-shipped}"
    timeout -k 10 120 python tools/diag/mb_conv.py wide 20 2>&1 | grep "^conv"
  done
done
