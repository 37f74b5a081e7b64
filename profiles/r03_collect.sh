#!/bin/bash
# Round 3: kernel-trace stats + PMC passes of the SAME bench command (run on the GPU box from the repo root; counters in
# their own runs, never combined with --kernel-trace / --stats), then the configuration-4 and e4m3 (configuration 5)
# records.  Output: gpurun_out/prof_r03/ (copy what is to be judged into profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/prof_r03
rm -rf $P; mkdir -p $P
BENCH="python3 bench.py --no-cpu-baseline --no-dropin"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- $BENCH --steps 10 --warmup 4 > $P/trace.log 2>&1
f=$(ls $P/trace/*/*kernel_stats.csv | head -1); cp $f $P/kernel_stats.csv
echo "trace done" >> $P/progress.log
# ONE stream: per-kernel durations of kernels that run alone (what the bench line's mfma_layers / hbm_layers quote)
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace1 -- $BENCH --streams 1 --steps 10 --warmup 4 > $P/trace1.log 2>&1
f1=$(ls $P/trace1/*/*kernel_stats.csv | head -1); cp $f1 $P/kernel_stats_1stream.csv
echo "trace1 done" >> $P/progress.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- $BENCH --steps 4 --warmup 2 > $P/fetch.log 2>&1
echo "fetch done" >> $P/progress.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- $BENCH --steps 4 --warmup 2 > $P/write.log 2>&1
echo "write done" >> $P/progress.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/mfma -- $BENCH --steps 4 --warmup 2 > $P/mfma.log 2>&1
echo "mfma done" >> $P/progress.log
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $P/rdreq -- $BENCH --steps 4 --warmup 2 > $P/rdreq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/l2 -- $BENCH --steps 4 --warmup 2 > $P/l2.log 2>&1
echo "pmc done" >> $P/progress.log
python3 - <<'PY'
import csv, glob, collections, json
P = 'gpurun_out/prof_r03'
def agg(d):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('%s/%s/*/*counter_collection.csv' % (P, d)):
        for r in csv.DictReader(open(f)):
            a[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return a
out = {}
for d in ('fetch', 'write', 'mfma', 'rdreq', 'l2'):
    for k, cs in agg(d).items():
        if 'sfvos' not in k: continue
        name = k.replace('sfvos::', '').split('(')[0].replace('void ', '')
        for c, v in cs.items():
            out.setdefault(name, {})[c] = {'launches': len(v), 'mean': sum(v) / len(v), 'max': max(v)}
json.dump(out, open(P + '/pmc_summary.json', 'w'), indent=1, sort_keys=True)
for name in sorted(out, key=lambda n: -out[n].get('FETCH_SIZE', {}).get('max', 0))[:8]:
    print(name[:60], {c: '%.4g' % v['max'] for c, v in out[name].items()})
PY
$BENCH --steps 20 --warmup 5 > $P/bench_line.json 2> $P/bench_line.err; tail -c 300 $P/bench_line.json
# configuration 4: (sp, fp) = (4, 64)
python3 bench.py --sp 4 --fp 64 --steps 10 --warmup 3 --no-cpu-baseline > $P/c4_bench_line.json 2> $P/c4_bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/c4trace -- python3 bench.py --sp 4 --fp 64 --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-dropin > $P/c4trace.log 2>&1
f=$(ls $P/c4trace/*/*kernel_stats.csv | head -1); cp $f $P/c4_kernel_stats_1stream.csv
echo "c4 done" >> $P/progress.log
# configuration 5: e4m3 conv path, eval forward
python3 bench.py --precision fp8 --steps 20 --warmup 5 > $P/fp8_bench_line.json 2> $P/fp8_bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/fp8trace -- python3 bench.py --precision fp8 --streams 1 --steps 10 --warmup 3 > $P/fp8trace.log 2>&1
f=$(ls $P/fp8trace/*/*kernel_stats.csv | head -1); cp $f $P/fp8_kernel_stats_1stream.csv
echo "fp8 done" >> $P/progress.log
# the RCCL path on one rank
python3 bench.py --no-cpu-baseline --no-dropin --no-layer-table --force-dist > $P/dist1_bench_line.json 2> $P/dist1_bench_line.err
rm -rf $P/trace $P/trace1 $P/fetch $P/write $P/mfma $P/rdreq $P/l2 $P/c4trace $P/fp8trace
ls -la $P
