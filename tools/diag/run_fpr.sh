#!/bin/bash
# bf16 3x3 weight gradients on 16x16x32 MFMAs -- frame pairs (fast_conv1) and row pairs (the others), shipped -- against
# the 32x32x16 configurations (SFVOS_WGRAD_M32) and against row pairs everywhere (SFVOS_WGRAD_RPR), switches of the
# diagnostic library: tests, fuzz, then interleaved micro-benchmarks
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad or grouped or window" > gpurun_out/fpr_tests.log 2>&1 || { tail -30 gpurun_out/fpr_tests.log; exit 1; }
tail -2 gpurun_out/fpr_tests.log
timeout -k 10 300 python -u tools/diag/fuzz_wgrad.py 120 9 > gpurun_out/fpr_fuzz.log 2>&1 || { tail -30 gpurun_out/fpr_fuzz.log; exit 1; }
tail -2 gpurun_out/fpr_fuzz.log
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2 3; do
  echo "== shipped (frame pairs f1, row pairs others)"; timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep "^wgrad" | grep " f1 \| s[123] \| f2 "
  echo "== 32x32x16"; SFVOS_WGRAD_M32=1 timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep "^wgrad" | grep " f1 \| s[123] \| f2 "
  echo "== row pairs everywhere"; SFVOS_WGRAD_RPR=1 timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep "^wgrad" | grep " f1 "
done
