# A/B: fast_conv1 forward launched first and alone (default) against the side stream starting with it (--overlap-f1), interleaved
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in "" "--overlap-f1"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-dropin --no-layer-table --steps 30 --warmup 6 $v > gpurun_out/ov.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/ov.json')); print('$v'.ljust(14), d['ms_per_step'], d['roofline']['launch_ms'])"
  done
done
