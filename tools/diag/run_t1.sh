#!/bin/bash
# wgrad_t1 kernel: tests, then warm / cold micro-benchmark against the generic kernel (diagnostic library switch).
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad" 2>&1 | tail -4 || exit 1
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2; do
  echo "== t1 kernel (cold)"; MB_COLD=1 timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 2>&1 | grep "f3"
  echo "== generic (cold)"; SFVOS_NO_T1_KERNEL=1 MB_COLD=1 timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 2>&1 | grep "f3"
done
echo "== t1 kernel (warm)"; timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 2>&1 | grep "f3"
for sh in 16 21; do echo "== shares $sh (cold)"; SFVOS_T1_SHARES=$sh MB_COLD=1 timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 2>&1 | grep " f3"; done
