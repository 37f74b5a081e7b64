#!/bin/bash
# frame-paired weight gradients (16x16x32 MFMAs, shipped) against the 32x32x16 configurations (SFVOS_WGRAD_M32 of the
# diagnostic library): tests, fuzz, then interleaved micro-benchmarks
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad or grouped or window" > gpurun_out/fpr_tests.log 2>&1 || { tail -30 gpurun_out/fpr_tests.log; exit 1; }
tail -2 gpurun_out/fpr_tests.log
timeout -k 10 300 python -u tools/diag/fuzz_wgrad.py 60 21 > gpurun_out/fpr_fuzz.log 2>&1 || { tail -30 gpurun_out/fpr_fuzz.log; exit 1; }
tail -2 gpurun_out/fpr_fuzz.log
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2 3; do
  echo "== frame pairs 16x16x32"; timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep "^wgrad" | grep " f1 \| s[123] \| f2 "
  echo "== 32x32x16"; SFVOS_WGRAD_M32=1 timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep "^wgrad" | grep " f1 \| s[123] \| f2 "
done
