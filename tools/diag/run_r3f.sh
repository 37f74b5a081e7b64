#!/bin/bash
cd $GRAFT_REPO_ROOT
python -u -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/t5.log 2>&1; echo "exit $?" >> gpurun_out/t5.log
tail -4 gpurun_out/t5.log
C=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
OUT=gpurun_out/ab_tall.log; : > $OUT
for i in 1 2 3; do
  for L in libsfvos.so libsfvos_notall.so; do
    echo "== $L" >> $OUT
    SFVOS_LIB=$C/$L timeout -k 10 120 python tools/diag/mb_conv.py f1 10 >> $OUT 2>&1 || exit 1
    SFVOS_LIB=$C/$L timeout -k 10 120 python tools/diag/mb_conv.py f2 10 >> $OUT 2>&1
    SFVOS_LIB=$C/$L timeout -k 10 120 python tools/diag/mb_conv.py df2 10 >> $OUT 2>&1
  done
done
grep -v amdgpu.ids $OUT
