#!/bin/bash
# the bench step with the 16x16x32 weight gradients (shipped: frame pairs / row pairs) against the 32x32x16 form
# (SFVOS_WGRAD_M32, diagnostic library), interleaved on one box; arguments: extra bench.py flags (e.g. --sp 4 --fp 64)
cd $GRAFT_REPO_ROOT
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2 3; do
  for v in m16 m32; do
    if [ $v = m32 ]; then export SFVOS_WGRAD_M32=1; else unset SFVOS_WGRAD_M32; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-dropin --steps 20 --warmup 4 "$@" > gpurun_out/fb_$v$i.json 2> gpurun_out/fb_$v$i.err
    python - <<P
import json
d=json.load(open('gpurun_out/fb_$v$i.json'))
k=d['kernels_ms']
print('$v', 'ms/step', d['ms_per_step'], 'f1 fwd', k['conv_fwd/f1'][1], 'f1 wgrad', k['wgrad/f1'][1], 's1 wgrad', k['wgrad/s1'][1], 's2 wgrad', k['wgrad/s2'][1], 's3 wgrad', k['wgrad/s3'][1])
P
  done
done
