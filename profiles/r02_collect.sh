#!/bin/bash
# Round 2: kernel-trace stats + PMC passes of the SAME bench command (run on the GPU box from the repo root).
# Counters in their own runs (no --kernel-trace / --stats combined with --pmc).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/prof_r02
rm -rf $P; mkdir -p $P
BENCH="python3 bench.py --no-cpu-baseline --no-dropin"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- $BENCH --steps 10 --warmup 4 > $P/trace.log 2>&1
# the same with ONE stream: per-kernel durations of kernels that run alone (with two streams a small kernel queued behind
# the other stream's full-chip kernel waits for a free CU, and rocprof counts that wait as its duration)
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace1 -- $BENCH --streams 1 --steps 10 --warmup 4 > $P/trace1.log 2>&1
f1=$(ls $P/trace1/*/*kernel_stats.csv | head -1); cp $f1 $P/kernel_stats_1stream.csv
rocprofv3 -L > $P/counters.txt 2>&1
grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*" $P/counters.txt | sort -u > $P/tcc_names.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- $BENCH --steps 4 --warmup 2 > $P/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- $BENCH --steps 4 --warmup 2 > $P/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/mfma -- $BENCH --steps 4 --warmup 2 > $P/mfma.log 2>&1
# request sizes at the fabric side of L2: total read requests and the 32-byte ones (the rest are 64-byte; gfx950 issues
# 128-byte reads as ONE request that FETCH_SIZE tallies at 64 B -- TCC_BUBBLE / 128B counters where the build has them)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $P/rdreq -- $BENCH --steps 4 --warmup 2 > $P/rdreq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/l2 -- $BENCH --steps 4 --warmup 2 > $P/l2.log 2>&1
if grep -q "TCC_EA0_RDREQ_128B" $P/tcc_names.txt; then
  rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum --output-format csv -d $P/rd128 -- $BENCH --steps 4 --warmup 2 > $P/rd128.log 2>&1
fi
if grep -q "TCC_EA0_RDREQ_DRAM" $P/tcc_names.txt; then
  rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $P/rddram -- $BENCH --steps 4 --warmup 2 > $P/rddram.log 2>&1
fi
python3 - <<'PY'
import csv, glob, collections, json
P = 'gpurun_out/prof_r02'
def agg(d):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('%s/%s/*/*counter_collection.csv' % (P, d)):
        for r in csv.DictReader(open(f)):
            a[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return a
out = {}
for d in ('fetch', 'write', 'mfma', 'rdreq', 'l2', 'rd128', 'rddram'):
    for k, cs in agg(d).items():
        if 'sfvos' not in k: continue
        name = k.replace('sfvos::', '').split('(')[0].replace('void ', '')
        for c, v in cs.items():
            out.setdefault(name, {})[c] = {'launches': len(v), 'mean': sum(v) / len(v), 'max': max(v)}
json.dump(out, open(P + '/pmc_summary.json', 'w'), indent=1, sort_keys=True)
for name in sorted(out, key=lambda n: -out[n].get('FETCH_SIZE', {}).get('max', 0))[:8]:
    print(name[:60], {c: '%.4g' % v['max'] for c, v in out[name].items()})
PY
f=$(ls $P/trace/*/*kernel_stats.csv | head -1); cp $f $P/kernel_stats.csv; head -14 $f | cut -c1-170
$BENCH --steps 20 --warmup 5 > $P/bench_line.json 2> $P/bench_line.err; tail -c 400 $P/bench_line.json
