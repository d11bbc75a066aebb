#!/bin/bash
# Round-6 evidence on the GPU box, on the FINAL library (one gpurun call from the repo root):
#   gpurun --timeout 2700 -- 'bash tools/collect_profiles_r06.sh'
# Output: gpurun_out/prof6/  (copy what is to be judged into profiles/ as r06_*)
#   bench.json                       the bench line of the library default (window form of the marching SpMV, blocks of 20 on the
#                                    matrix-instruction kernels, fused rotation, speculative chain), with cpu_baseline and shift_invert
#   bench_stencil2.json              KS_STENCIL_MARCH=0: round 5's SpMV kernel, everything else the same (A/B of the round's kernel work)
#   bench_march_registers.json       KS_MARCH_WINDOW=0: the marching kernel with every tap through registers
#   bench_sstep0.json                the per-step expansion
#   bench_under_rocprof.json, kernel_stats.csv      rocprofv3 --kernel-trace --stats of the default run
#   pmc_sstep20_{fetch,write}.csv, pmc_summary_sstep20.txt, pmc_traffic.json   --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes
#                                    (22 timed cycles: >= 20 launches of every block class) -- pmc_traffic.json RE-COLLECTED on this library
#   cfg{2,3,4}_sstep{0,20}.json, cfg*_kernel_stats.csv   BASELINE configs 2-4 (tools/config_bench.py), both forms
#   full_solves.txt                  whole solves to convergence (tools/full_solve_check.py): 216^3 tol 1e-6, 100^3 tol 1e-8
#   dist_overhead.txt                fixed cost of the multi-GPU structure at the 8-way share of 216^3 (tools/dist_overhead.py 108)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof6
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
B="python $REPO/bench.py --no-cpu-baseline --no-shift-invert"
python $REPO/bench.py --steps 20 > $OUT/bench.json 2> $OUT/bench.err; check $OUT/bench.err
KS_STENCIL_MARCH=0 $B --steps 20 > $OUT/bench_stencil2.json 2>> $OUT/bench.err
KS_MARCH_WINDOW=0 $B --steps 20 > $OUT/bench_march_registers.json 2>> $OUT/bench.err
$B --sstep 0 > $OUT/bench_sstep0.json 2>> $OUT/bench.err
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
  for s in 0 20; do
    python $REPO/tools/config_bench.py $cfg --sstep $s > $OUT/${cfg}_sstep${s}.json 2> $OUT/${cfg}.err
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$cfg -- python $REPO/tools/config_bench.py $cfg --sstep 20 > /dev/null 2>> $OUT/${cfg}.err
  cp "$(find /tmp/kt_$cfg -name '*kernel_stats.csv' | head -1)" $OUT/${cfg}_kernel_stats.csv
done
cd $REPO
{ python tools/full_solve_check.py 216 20 1e-6; python tools/full_solve_check.py 100 20 1e-8; } > $OUT/full_solves.txt 2>&1
{ echo "# tools/dist_overhead.py 108: the 8-way share of 216^3 on ONE GPU, one restart cycle per call (ms per Arnoldi iteration), separate processes";
  for leg in plain rccl p2p plain rccl p2p; do python tools/dist_overhead.py 108 $leg 2>&1 | grep ms/iter; done;
  echo "# the whole 216^3 on the plain context, same protocol"; python tools/dist_overhead.py 216 plain 2>&1 | grep ms/iter; } > $OUT/dist_overhead.txt 2>&1
ls -la $OUT
if [ $failed -ne 0 ]; then echo "# RESULT: $failed record(s) FAILED"; exit 1; fi
grep -l Traceback $OUT/*.err && echo "# RESULT: some record FAILED" || echo "# RESULT: all records collected"
