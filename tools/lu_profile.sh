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
