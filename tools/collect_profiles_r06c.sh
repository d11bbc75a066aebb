#!/bin/bash
# Round-6 evidence, second collection (r06c_*): on the library whose restart waits for an EVENT instead of the stream and with the
# timed cycles of bench.py / config_bench.py back to back (one gpurun call from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles_r06c.sh'
# Output: gpurun_out/prof6c/  (copied into profiles/ as r06c_*)
#   bench.json                 the bench line of the library default, with cpu_baseline and shift_invert
#   bench_sync_cycles.json     --sync-cycles: every timed cycle synchronised (how all earlier records were taken)
#   bench_stream_sync.json     KS_QSTAGE_EVENT=0: the restart synchronises the stream as before (A/B of the change)
#   bench_under_rocprof.json, kernel_stats.csv        rocprofv3 --kernel-trace --stats of the default run
#   pmc_sstep20_{fetch,write}.csv, pmc_summary_sstep20.txt, pmc_traffic.json   --pmc FETCH_SIZE / WRITE_SIZE, SEPARATE passes
#   cfg{2,3,4}_sstep20.json, cfg3_true_start.json     BASELINE configs 2-4 (tools/config_bench.py); config 3 with KS_TRUE_START=1
#   full_solves.txt            whole solves to convergence (tools/full_solve_check.py)
#   outlier_solves.txt         :LM problems with dominant outliers: in-chain deflation on / off / step by step (tools/outlier_solves.py)
#   dist_overhead.txt          tools/dist_overhead.py 108 (8-way share of 216^3 on one GPU), three transports, and the whole problem
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof6c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
trim() {
  python - "$1" "$2" <<PY
import csv, sys
r = csv.DictReader(open(sys.argv[1]))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
for x in r:
    w.writerow([x["Dispatch_Id"], x["Kernel_Name"], x["Counter_Name"], x["Counter_Value"]])
PY
}
B="python $REPO/bench.py --no-cpu-baseline --no-shift-invert"
python $REPO/bench.py --steps 20 > $OUT/bench.json 2> $OUT/bench.err
$B --steps 20 --sync-cycles > $OUT/bench_sync_cycles.json 2>> $OUT/bench.err
KS_QSTAGE_EVENT=0 $B --steps 20 > $OUT/bench_stream_sync.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $B --steps 20 > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
N=10077696; NNZ=70263936
cp $REPO/profiles/pmc_traffic.json $OUT/pmc_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_20_$c -- $B --steps 22 --warmup 2 --no-profile > /dev/null 2>> $OUT/bench.err
  trim "$(find /tmp/pmc_20_$c -name '*counter_collection.csv' | head -1)" "$OUT/pmc_sstep20_$(echo $c | tr A-Z a-z | sed s/_size//).csv"
done
python $REPO/tools/pmc_summary.py $OUT/pmc_sstep20_fetch.csv $OUT/pmc_sstep20_write.csv $N $NNZ $OUT/pmc_traffic.json 0.1434 0 62 21 > $OUT/pmc_summary_sstep20.txt 2>&1
for cfg in cfg2 cfg3 cfg4; do
  python $REPO/tools/config_bench.py $cfg --sstep 20 --steps 20 > $OUT/${cfg}_sstep20.json 2> $OUT/${cfg}.err
done
KS_TRUE_START=1 python $REPO/tools/config_bench.py cfg3 --sstep 20 --steps 20 > $OUT/cfg3_true_start.json 2>> $OUT/cfg3.err
cd $REPO
{ python tools/full_solve_check.py 216 20 1e-6; python tools/full_solve_check.py 100 20 1e-8; } > $OUT/full_solves.txt 2>&1
python tools/outlier_solves.py > $OUT/outlier_solves.txt 2>&1
{ echo "# tools/dist_overhead.py 108: the 8-way share of 216^3 on ONE GPU, cycles back to back (ms per Arnoldi iteration), separate processes";
  for leg in plain rccl p2p plain rccl p2p; do python tools/dist_overhead.py 108 $leg 2>&1 | grep ms/iter; done;
  echo "# the whole 216^3 on the plain context, same protocol"; python tools/dist_overhead.py 216 plain 2>&1 | grep ms/iter; } > $OUT/dist_overhead.txt 2>&1
ls -la $OUT
grep -l Traceback $OUT/*.err && echo "# RESULT: some record FAILED" || echo "# RESULT: all records collected"
