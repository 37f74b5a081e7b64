#!/bin/bash
# persistent frame-split kernel: tests, then micro-benchmarks against one workgroup per item (same code, diagnostic switch)
# and against the library built from the previous conv3d.hip
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv3d or grouped or fp8" 2>&1 | tail -3 || exit 1
for i in 1 2; do
  for v in new nonpersist old; do
    unset SFVOS_FS_NONPERSIST
    if [ $v = old ]; then export SFVOS_LIB=$L/libsfvos_old.so; else export SFVOS_LIB=$L/libsfvos_diag.so; fi
    if [ $v = nonpersist ]; then export SFVOS_FS_NONPERSIST=1; fi
    echo "== $v"
    timeout -k 10 120 python tools/diag/mb_conv.py f1 10 2>&1 | grep "^conv"
    timeout -k 10 120 python tools/diag/mb_conv.py f2 20 2>&1 | grep "^conv"
    timeout -k 10 120 python tools/diag/mb_conv.py df2 20 2>&1 | grep "^conv"
  done
done
