#!/bin/bash
# full GPU test suite, then the four bench lines
cd $GRAFT_REPO_ROOT
python -u -m pytest tests -m gpu -q > gpurun_out/t_full.log 2>&1; echo "exit $?" >> gpurun_out/t_full.log
tail -4 gpurun_out/t_full.log
python -u bench.py --no-cpu-baseline > gpurun_out/b_c2.json 2> gpurun_out/b_c2.err; echo "bench c2 $?"
python -u bench.py --no-cpu-baseline --sp 4 --fp 64 --steps 10 --warmup 2 > gpurun_out/b_c4.json 2> gpurun_out/b_c4.err; echo "bench c4 $?"
