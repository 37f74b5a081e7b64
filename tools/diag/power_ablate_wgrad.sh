#!/bin/bash
# What draws the power in fast_conv1's weight gradient (frame-paired 16x16x32 stages)?  Timing-only builds
# (-DSFVOS_WG_ABLATE=1: x fragments read PF + 1 times per stage, 2: dy fragments three times, 3: both, 4: no staging copies
# inside the stages; libsfvos_wabl<N>.so = libsfvos.so with wgrad.hip compiled that way), power and clock polled beside
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
poll() { for i in $(seq 1 $2); do p=$(rocm-smi -d 0 --showpower 2>/dev/null | grep "Power (W)" | sed 's/.*: //'); c=$(rocm-smi -d 0 --showclocks 2>/dev/null | grep -i "sclk" | grep -o "([0-9]*Mhz)" | head -1); echo "  $1 power=$p sclk=$c"; sleep 1; done; }
run() { lab=$1; shift
  env "$@" timeout -k 10 60 python tools/diag/mb_conv.py wf1 2500 > gpurun_out/pw_$lab.log 2>&1 &
  sleep 5; poll $lab 2; wait
  grep "^wgrad" gpurun_out/pw_$lab.log
}
for i in 1 2; do
  run shipped X=1
  run x_frags_once SFVOS_LIB=$L/libsfvos_wabl1.so
  run dy_frags_once SFVOS_LIB=$L/libsfvos_wabl2.so
  run no_rereads SFVOS_LIB=$L/libsfvos_wabl3.so
  run no_stage_copies SFVOS_LIB=$L/libsfvos_wabl4.so
done
