#!/bin/bash
# GraphedStep: tests, then eager vs replayed cycle on the reference's default (1,1), on (3,7) and on the headline (4,32)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_graph.py -m gpu -q -x -s 2>&1 | tail -12 || exit 1
for cfg in "1 1" "3 7" "4 32"; do
  set -- $cfg
  for g in "" "--graph"; do
    timeout -k 10 300 python bench.py --sp $1 --fp $2 $g --no-cpu-baseline --no-dropin --steps 20 --warmup 4 > gpurun_out/b_g.json 2>gpurun_out/b_g.err || { echo "($1,$2) $g FAILED"; tail -5 gpurun_out/b_g.err; continue; }
    python - <<PY
import json
d=json.loads(open('gpurun_out/b_g.json').read().strip().splitlines()[-1])
print('($1,$2) ${g:-eager}', d['value'], 'clips/s', d['ms_per_step'], 'ms', 'roofline', d['roofline']['frac'])
PY
  done
done
