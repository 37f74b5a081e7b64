#!/bin/bash
# round-3 check: new kernels' tests, then the bench lines (headline, C4, fp8, RCCL on one rank)
set -o pipefail
mkdir -p gpurun_out
python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_mask_head.py -m gpu -q -x > gpurun_out/t2.log 2>&1; echo "exit $?" >> gpurun_out/t2.log
tail -3 gpurun_out/t2.log
python -u -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "mask_flips or config4" > gpurun_out/t3.log 2>&1; echo "exit $?" >> gpurun_out/t3.log
tail -3 gpurun_out/t3.log
python -u bench.py --no-cpu-baseline > gpurun_out/b_c2.json 2> gpurun_out/b_c2.err; echo "bench c2 $?"
python -u bench.py --no-cpu-baseline --sp 4 --fp 64 --steps 10 --warmup 2 > gpurun_out/b_c4.json 2> gpurun_out/b_c4.err; echo "bench c4 $?"
python -u bench.py --precision fp8 > gpurun_out/b_fp8.json 2> gpurun_out/b_fp8.err; echo "bench fp8 $?"
python -u bench.py --no-cpu-baseline --no-layer-table --no-dropin --force-dist > gpurun_out/b_dist.json 2> gpurun_out/b_dist.err; echo "bench dist $?"
