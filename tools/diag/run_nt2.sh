#!/bin/bash
cd $GRAFT_REPO_ROOT
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2; do
for m in 128 192; do
  SFVOS_NT2_MIN=$m timeout -k 10 300 python bench.py --sp 1 --fp 1 --graph --no-cpu-baseline --no-dropin --steps 40 --warmup 4 > gpurun_out/b_g.json 2>gpurun_out/b_g.err || { echo FAILED; tail -5 gpurun_out/b_g.err; continue; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/b_g.json').read().strip().splitlines()[-1])
print('nt2_min $m', d['value'], 'clips/s', d['ms_per_step'], 'ms', {k:v[1] for k,v in d['kernels_ms'].items() if 'conv_fwd/s' in k})
PY
done
done
