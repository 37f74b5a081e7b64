#!/bin/bash
# weight gradients with v_mfma_f32_16x16x32_bf16 on 32-pixel tiles against the 32x32x16 configurations on 16-pixel tiles
# (SFVOS_WGRAD_M32 of a diagnostic library), for x-fragment read-ahead depths PF = 1, 2, 3 (libsfvos_pf<N>.so: diagnostic
# builds with -DSFVOS_WGRAD_PF=N): interleaved micro-benchmarks
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
for i in 1 2; do
  for pf in 1 2 3; do
    echo "== 16x16x32 PF=$pf"; SFVOS_LIB=$L/libsfvos_pf$pf.so timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep "^wgrad" | grep " f1 \| s[123] \| f2 "
  done
  echo "== 32x32x16"; SFVOS_LIB=$L/libsfvos_pf3.so SFVOS_WGRAD_M32=1 timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep "^wgrad" | grep " f1 \| s[123] \| f2 "
done
