#!/bin/bash
# frame-split forward convs on 16-pixel tiles with blocks of 8 frames (shipped) against 32-pixel tiles with blocks of 4
# (SFVOS_FS_TW32, diagnostic library): tests, fuzz, interleaved micro-benchmarks of fast_conv1 / fast_conv2 forward
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv3d or grouped or window or fp8" > gpurun_out/fs16_tests.log 2>&1 || { tail -30 gpurun_out/fs16_tests.log; exit 1; }
tail -2 gpurun_out/fs16_tests.log
timeout -k 10 300 python -u tools/diag/fuzz_conv.py 80 31 > gpurun_out/fs16_fuzz.log 2>&1 || { tail -30 gpurun_out/fs16_fuzz.log; exit 1; }
tail -2 gpurun_out/fs16_fuzz.log
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2 3; do
  echo "== 16-px tiles, blocks of 8"; for w in f1 f2; do timeout -k 10 120 python tools/diag/mb_conv.py $w 20 2>&1 | grep "^conv"; done
  echo "== 32-px tiles, blocks of 4"; for w in f1 f2; do SFVOS_FS_TW32=1 timeout -k 10 120 python tools/diag/mb_conv.py $w 20 2>&1 | grep "^conv"; done
done
