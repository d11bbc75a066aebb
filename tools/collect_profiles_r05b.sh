#!/bin/bash
# Round-5 evidence, second collection (after: block sizes at run time, ComplexF64 fused rotation + speculative chain, split
# rotation for uninstantiated shapes).  One gpurun call from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles_r05b.sh'
# Output: gpurun_out/prof5b/ (copy what is to be judged into profiles/ as r05b_*)
#   bench.json                         the bench line of the final library (with cpu_baseline and shift_invert)
#   bench_under_rocprof.json, kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default run
#   cfg{2,3,4}_sstep{0,20}.json, cfg{3,4}_kernel_stats.csv   BASELINE configs 2-4 (tools/config_bench.py), both forms
#   full_solves.txt                    whole solves to convergence (tools/full_solve_check.py): 216^3 tol 1e-6, 100^3 tol 1e-8
#   dist_overhead.txt                  fixed cost of the multi-GPU structure at the 8-way share of 216^3 (tools/dist_overhead.py 108)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof5b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-shift-invert"
python $REPO/bench.py --steps 20 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $B --steps 20 > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
for cfg in cfg2 cfg3 cfg4; do
  for s in 0 20; do
    python $REPO/tools/config_bench.py $cfg --sstep $s > $OUT/${cfg}_sstep${s}.json 2> $OUT/${cfg}.err
  done
done
for cfg in cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$cfg -- python $REPO/tools/config_bench.py $cfg --sstep 20 > /dev/null 2>> $OUT/${cfg}.err
  cp "$(find /tmp/kt_$cfg -name '*kernel_stats.csv' | head -1)" $OUT/${cfg}_kernel_stats.csv
done
cd $REPO
{ python tools/full_solve_check.py 216 20 1e-6; python tools/full_solve_check.py 100 20 1e-8; } > $OUT/full_solves.txt 2>&1
{ echo "# tools/dist_overhead.py 108: the 8-way share of 216^3 on ONE GPU, one restart cycle per call (ms per Arnoldi iteration), separate processes";
  for leg in plain rccl p2p plain rccl p2p; do python tools/dist_overhead.py 108 $leg 2>&1 | grep ms/iter; done;
  echo "# the whole 216^3 on the plain context, same protocol"; python tools/dist_overhead.py 216 plain 2>&1 | grep ms/iter; } > $OUT/dist_overhead.txt 2>&1
ls -la $OUT
grep -l Traceback $OUT/*.err && echo "# RESULT: some record FAILED" || echo "# RESULT: all records collected"
