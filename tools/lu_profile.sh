#!/bin/bash
# only the shift-invert part of tools/collect_profiles.sh (same output files)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/lu_bench.py 200 250 --reps 20 --solve > $OUT/lu_n5e4.txt 2>&1
LU_RESTARTS=6 python $REPO/tools/lu_bench.py 500 1000 --reps 10 --solve > $OUT/lu_n5e5.txt 2>&1
KS_LU_STATS=1 python $REPO/tools/lu_bench.py 500 1000 --reps 1 2>&1 | grep "lu L\|lu U" | tail -2 >> $OUT/lu_n5e5.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_lu -- python $REPO/tools/lu_bench.py 500 1000 --reps 20 > /dev/null 2>> $OUT/lu.err
cp "$(find /tmp/kt_lu -name '*kernel_stats.csv' | head -1)" $OUT/lu_kernel_stats.csv
# HBM traffic of the solve kernels: two separate --pmc passes (kernel trace only), summarised by tools/pmc_generic.py
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
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_lu_$c -- python $REPO/tools/lu_bench.py 500 1000 --reps 4 > /dev/null 2>> $OUT/lu.err
  trim "$(find /tmp/pmc_lu_$c -name '*counter_collection.csv' | head -1)" "$OUT/lu_pmc_$(echo $c | tr A-Z a-z | sed s/_size//).csv"
done
python $REPO/tools/pmc_generic.py $OUT/lu_pmc_fetch.csv $OUT/lu_pmc_write.csv > $OUT/lu_pmc_summary.txt 2>&1
