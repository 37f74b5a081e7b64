#!/bin/bash
# kernel trace of the bench step on ONE stream (every kernel alone): per-kernel durations in situ
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_step
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_step -o step -- python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --no-cpu-baseline --no-dropin --no-layer-table --steps 8 --warmup 2 $@ > $GRAFT_REPO_ROOT/gpurun_out/prof_step.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_step.log | cut -c1-300
