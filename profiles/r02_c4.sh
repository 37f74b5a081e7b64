#!/bin/bash
# BASELINE config 4 ("T=64, alpha=16" = (sp, fp) = (4, 64)): bench line + kernel-trace stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/prof_r02_c4
rm -rf $P; mkdir -p $P
python3 bench.py --sp 4 --fp 64 --steps 10 --warmup 3 --no-cpu-baseline > $P/bench_line.json 2> $P/bench_line.err
tail -c 600 $P/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --sp 4 --fp 64 --steps 6 --warmup 2 --no-cpu-baseline --no-dropin > $P/trace.log 2>&1
f=$(ls $P/trace/*/*kernel_stats.csv | head -1); cp $f $P/kernel_stats.csv; head -12 $f | cut -c1-170
