#!/bin/bash
# What draws the power in fast_conv1's forward?  The kernel in timing-only builds (wrong results) that read the pixel
# fragments (libsfvos_abl1.so), the weight fragments (abl2) or both (abl3) only once per stage (-DSFVOS_FS_ABLATE=1/2/3),
# and the shipped kernel without its staging copies after the first stage (SFVOS_CONV_DEBUG=2, diagnostic library):
# time per launch with board power and shader clock polled beside.  Build the three libraries first:
#   cd csrc; for v in 1 2 3; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSFVOS_FS_ABLATE=$v -c conv3d.hip -o /tmp/c$v.o &&
#     hipcc --offload-arch=gfx950 -shared -fPIC -o libsfvos_abl$v.so /tmp/c$v.o $(ls *.o | grep -v 'diag\|conv3d'); done
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
poll() { for i in $(seq 1 $2); do p=$(rocm-smi -d 0 --showpower 2>/dev/null | grep "Power (W)" | sed 's/.*: //'); c=$(rocm-smi -d 0 --showclocks 2>/dev/null | grep -i "sclk" | grep -o "([0-9]*Mhz)" | head -1); echo "  $1 power=$p sclk=$c"; sleep 1; done; }
run() {  # label, env...
  lab=$1; shift
  env "$@" timeout -k 10 60 python tools/diag/mb_conv.py f1 2500 > gpurun_out/pa_$lab.log 2>&1 &
  sleep 5; poll $lab 3; wait
  grep "^conv" gpurun_out/pa_$lab.log
}
for i in 1 2; do
  run shipped X=1
  run no_pixel_rereads SFVOS_LIB=$L/libsfvos_abl1.so
  run no_weight_rereads SFVOS_LIB=$L/libsfvos_abl2.so
  run no_rereads SFVOS_LIB=$L/libsfvos_abl3.so
  run no_copies SFVOS_LIB=$L/libsfvos_diag.so SFVOS_CONV_DEBUG=2
done
