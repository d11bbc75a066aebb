#!/bin/bash
# Round-4 evidence on the GPU box (one gpurun call from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles_r04.sh'
# Output: gpurun_out/prof4/  (copy what is to be judged into profiles/ as r04_*)
#   bench.json                          the bench line (s-step expansion, default --sstep 20), with cpu_baseline and shift_invert
#   bench_sstep0.json / bench_sstep5.json / bench_sstep8.json / bench_sstep10.json  the same workload step by step / in blocks of 5 / 8 / 10 (no CPU legs)
#   bench_under_rocprof.json, kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default run
#   pmc_fetch.csv, pmc_write.csv, pmc_summary.txt, pmc_traffic.json   --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (with
#                                       --kernel-trace only), block kernels; then the per-step kernels (sstep 0) merged in
#   cfg{2,3,4}_sstep{0,20}.json, cfg*_kernel_stats.csv   BASELINE configs 2-4 (tools/config_bench.py), both forms
#   blk_bench.txt                       the two streaming kernels stand-alone (tools/blk_bench.py)
#   fin_blk_timing.txt                  stage timers of the reduction + algebra kernel (rebuilds with -DKS_FIN_TIMING, then back)
# A record that contains a Traceback is renamed *.FAILED and the script exits non-zero.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
failed=0
check() { if grep -q "Traceback" "$1" 2>/dev/null; then mv "$1" "$1.FAILED"; echo "!! $1 contains a Traceback"; failed=$((failed+1)); fi; }
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
python $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err; check $OUT/bench.err
python $REPO/bench.py --no-cpu-baseline --no-shift-invert --sstep 0 > $OUT/bench_sstep0.json 2>> $OUT/bench.err
python $REPO/bench.py --no-cpu-baseline --no-shift-invert --sstep 5 > $OUT/bench_sstep5.json 2>> $OUT/bench.err
python $REPO/bench.py --no-cpu-baseline --no-shift-invert --sstep 8 > $OUT/bench_sstep8.json 2>> $OUT/bench.err
python $REPO/bench.py --no-cpu-baseline --no-shift-invert --sstep 10 > $OUT/bench_sstep10.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $REPO/bench.py --no-cpu-baseline --no-shift-invert > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
N=10077696; NNZ=70263936
rm -f $OUT/pmc_traffic.json
for mode in 0 20; do   # per-step kernels first, then the block kernels: the summary of the second pass keeps the classes of the first
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${mode}_$c -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-shift-invert --no-profile --sstep $mode > /dev/null 2>> $OUT/bench.err
    trim "$(find /tmp/pmc_${mode}_$c -name '*counter_collection.csv' | head -1)" "$OUT/pmc_sstep${mode}_$(echo $c | tr A-Z a-z | sed s/_size//).csv"
  done
  python $REPO/tools/pmc_summary.py $OUT/pmc_sstep${mode}_fetch.csv $OUT/pmc_sstep${mode}_write.csv $N $NNZ $OUT/pmc_traffic.json 0.1434 0 62 21 > $OUT/pmc_summary_sstep${mode}.txt 2>&1
done
for cfg in cfg2 cfg3 cfg4; do
  for s in 0 20; do
    python $REPO/tools/config_bench.py $cfg --sstep $s > $OUT/${cfg}_sstep${s}.json 2> $OUT/${cfg}.err
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$cfg -- python $REPO/tools/config_bench.py $cfg --sstep 20 > /dev/null 2>> $OUT/${cfg}.err
  cp "$(find /tmp/kt_$cfg -name '*kernel_stats.csv' | head -1)" $OUT/${cfg}_kernel_stats.csv
done
cd $REPO
python tools/blk_bench.py > $OUT/blk_bench.txt 2>&1; check $OUT/blk_bench.txt
cp arnoldimethod.jl_amd/libkschur_hip.so /tmp/libkschur_hip.so.keep
bash tools/fin_blk_timing.sh 216 > $OUT/fin_blk_timing.txt 2>&1
cp /tmp/libkschur_hip.so.keep arnoldimethod.jl_amd/libkschur_hip.so
ls -la $OUT
if [ $failed -ne 0 ]; then echo "# RESULT: $failed record(s) FAILED"; exit 1; fi
echo "# RESULT: all records collected"
