#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
# Output: gpurun_out/prof/ {bench.json, bench_under_rocprof.json, kernel_stats.csv, pmc_fetch.csv, pmc_write.csv}
# --pmc passes are separate runs with --kernel-trace only (gpurun refuses other combinations).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $REPO/bench.py > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>> $OUT/bench.err
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  # keep only what pmc_summary.py reads (the raw file is tens of MB)
  python - "$f" "$OUT/pmc_$(echo $c | tr A-Z a-z | sed s/_size//).csv" <<PY
import csv, sys
r = csv.DictReader(open(sys.argv[1]))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
for x in r:
    w.writerow([x["Dispatch_Id"], x["Kernel_Name"], x["Counter_Name"], x["Counter_Value"]])
PY
done
ls -la $OUT
