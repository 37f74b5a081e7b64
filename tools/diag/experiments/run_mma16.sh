#!/bin/bash
# TIMING ONLY: the weight-gradient and wide conv kernels with every v_mfma_f32_32x32x16_bf16 replaced by two independent
# v_mfma_f32_16x16x32_bf16 on the same operand registers (same FLOPs, same pipe cycles, wrong results): what the MFMA shape
# alone is worth inside the real kernels (board power and clock polled beside)
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
poll() { for i in $(seq 1 $2); do p=$(rocm-smi -d 0 --showpower 2>/dev/null | grep "Power (W)" | sed 's/.*: //'); c=$(rocm-smi -d 0 --showclocks 2>/dev/null | grep -i "sclk" | grep -o "([0-9]*Mhz)" | head -1); echo "  $1 power=$p sclk=$c"; sleep 1; done; }
for i in 1 2; do
  for v in shipped mma16; do
    if [ $v = mma16 ]; then export SFVOS_LIB=$L/libsfvos_mma16.so; else unset SFVOS_LIB; fi
    echo "== $v"
    timeout -k 10 120 python tools/diag/mb_conv.py wf1 1500 > gpurun_out/m16_$v.log 2>&1 &
    sleep 5; poll $v 2; wait
    grep "^wgrad" gpurun_out/m16_$v.log
    timeout -k 10 120 python tools/diag/mb_conv.py wide 20 2>&1 | grep "^conv"
    timeout -k 10 120 python tools/diag/mb_conv.py wall 20 2>&1 | grep "^wgrad" | grep " s[123] "
  done
done
