#!/bin/bash
# lateral weight-gradient kernel: workgroup count / ring depth sweep (diagnostic build), then a kernel trace of the default
cd $GRAFT_REPO_ROOT
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
OUT=gpurun_out/lwg_sweep.log; : > $OUT
for WGS in 64 96 128 160 192 256; do
  for RING in 3 4 5 6; do
    echo "== WGS $WGS RING $RING" >> $OUT
    SFVOS_LWG_WGS=$WGS SFVOS_LWG_RING=$RING timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 >> $OUT 2>&1 || exit 1
  done
done
unset SFVOS_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_wlat -o wlat -- python3 $GRAFT_REPO_ROOT/tools/diag/mb_conv.py wlat 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_wlat.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_wlat -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/wlat_kernel_stats.csv
tail -8 $OUT
