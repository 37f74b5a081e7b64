#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --no-cpu-baseline --no-dropin --no-layer-table --kernel-events none --steps 10 --warmup 4 > gpurun_out/prof_tl.log 2>&1
f=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
head -1 $f
python3 tools/diag/timeline.py $f > gpurun_out/timeline.txt 2>&1
cat gpurun_out/timeline.txt
