#!/bin/bash
# Board power and shader clock while the dominant kernel runs back to back (rocm-smi polled from the side), then the same
# launch on all-zero operands, then idle: is the MFMA rate set by cycles or by what the power limit lets the clock be?
cd $GRAFT_REPO_ROOT
poll() {  # $1 = label, $2 = seconds
  for i in $(seq 1 $2); do
    p=$(rocm-smi -d 0 --showpower 2>/dev/null | grep "Power (W)" | sed 's/.*: //')
    c=$(rocm-smi -d 0 --showclocks 2>/dev/null | grep -i "sclk" | grep -o "([0-9]*Mhz)" | head -1)
    m=$(rocm-smi -d 0 --showmaxpower 2>/dev/null | grep "Power (W)" | sed 's/.*: //')
    echo "$1 t=$i power=$p sclk=$c cap=$m"
    sleep 1
  done
}
rocm-smi -d 0 --showpower --showclocks --showmaxpower 2>&1 | head -30
poll idle 3
timeout -k 10 60 python tools/diag/mb_conv.py f1 4000 > gpurun_out/pc_rand.log 2>&1 &
sleep 6; poll f1_random 6; wait
cat gpurun_out/pc_rand.log | grep "^conv"
MB_FILL=zeros timeout -k 10 60 python tools/diag/mb_conv.py f1 4000 > gpurun_out/pc_zero.log 2>&1 &
sleep 6; poll f1_zeros 6; wait
cat gpurun_out/pc_zero.log | grep "^conv"
timeout -k 10 60 python tools/diag/mb_conv.py wf1 4000 > gpurun_out/pc_wf1.log 2>&1 &
sleep 6; poll wgrad_f1_random 6; wait
cat gpurun_out/pc_wf1.log | grep "^wgrad"
timeout -k 10 60 python tools/diag/mb_conv.py wide 2500 > gpurun_out/pc_wide.log 2>&1 &
sleep 6; poll wide_convs_random 6; wait
cat gpurun_out/pc_wide.log | grep "^conv"
python bench.py --no-cpu-baseline --no-dropin --no-layer-table --steps 1500 --warmup 4 > gpurun_out/pc_bench.json 2>/dev/null &
sleep 8; poll bench_step 6; wait
python -c "import json; d=json.loads(open('gpurun_out/pc_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
