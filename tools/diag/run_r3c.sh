#!/bin/bash
cd $GRAFT_REPO_ROOT
python -u -m pytest tests/test_gpu_roi_align.py tests/test_gpu_kernels.py -m gpu -q -x -k "roi or lateral or wgrad" > gpurun_out/t4.log 2>&1; echo "exit $?" >> gpurun_out/t4.log
tail -4 gpurun_out/t4.log
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
OUT=gpurun_out/lwg_sweep3.log; : > $OUT
for SUB in 1 2; do
for WGS in 128 192 256; do
  for RING in 2 3 5; do
    echo "== WGS $WGS RING $RING SUB $SUB" >> $OUT
    SFVOS_LWG_SUB=$SUB SFVOS_LWG_WGS=$WGS SFVOS_LWG_RING=$RING timeout -k 10 120 python tools/diag/mb_conv.py wlat 20 >> $OUT 2>&1 || exit 1
  done
done
done
unset SFVOS_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_wlat3 -o wlat -- python3 $GRAFT_REPO_ROOT/tools/diag/mb_conv.py wlat 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_wlat3.log 2>&1
