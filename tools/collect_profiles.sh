#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh'
# Output: gpurun_out/prof/ {bench.json, bench_under_rocprof.json, kernel_stats.csv, pmc_fetch.csv, pmc_write.csv} for
# the headline (bench.py) and <cfg>_{bench.json,kernel_stats.csv,pmc_fetch.csv,pmc_write.csv} for configs 3 and 4
# (tools/config_bench.py).  --pmc passes are separate runs with --kernel-trace only (gpurun refuses other combinations).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
trim() {  # keep only what the summarisers read (the raw file is tens of MB)
  python - "$1" "$2" <<PY
import csv, sys
r = csv.DictReader(open(sys.argv[1]))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
for x in r:
    w.writerow([x["Dispatch_Id"], x["Kernel_Name"], x["Counter_Name"], x["Counter_Value"]])
PY
}
python $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
KS_PLACE_TRIALS=2 KS_PLACE_CONTIGUOUS=1 python $REPO/bench.py --no-cpu-baseline > $OUT/bench_placement_contiguous.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>> $OUT/bench.err
  trim "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" "$OUT/pmc_$(echo $c | tr A-Z a-z | sed s/_size//).csv"
done
for cfg in cfg3 cfg4 cfg4big; do
  python $REPO/tools/config_bench.py $cfg > $OUT/${cfg}_bench.json 2> $OUT/${cfg}.err
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$cfg -- python $REPO/tools/config_bench.py $cfg > /dev/null 2>> $OUT/${cfg}.err
  cp "$(find /tmp/kt_$cfg -name '*kernel_stats.csv' | head -1)" $OUT/${cfg}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${cfg}_$c -- python $REPO/tools/config_bench.py $cfg --steps 3 --warmup 1 > /dev/null 2>> $OUT/${cfg}.err
    trim "$(find /tmp/pmc_${cfg}_$c -name '*counter_collection.csv' | head -1)" "$OUT/${cfg}_pmc_$(echo $c | tr A-Z a-z | sed s/_size//).csv"
  done
done
# round 3 additions
# (a) the SpMV north_star names: plain CSR through k_spmv_csr on the 216^3 Laplacian, traffic next to it
cd $REPO
SPMV_TRIALS=2 python tools/spmv_bench.py lap216 lap100 lapvar216 hashed1e6 skew1e6 cplx5e5 > $OUT/spmv_layouts.txt 2>> $OUT/bench.err
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  KS_SPMV_FORMAT=csr SPMV_TRIALS=1 SPMV_REPS=20 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_spmv_$c -- python $REPO/tools/spmv_bench.py lap216 > /dev/null 2>> $OUT/bench.err
  trim "$(find /tmp/pmc_spmv_$c -name '*counter_collection.csv' | head -1)" "$OUT/spmv_csr_pmc_$(echo $c | tr A-Z a-z | sed s/_size//).csv"
done
python $REPO/tools/pmc_generic.py $OUT/spmv_csr_pmc_fetch.csv $OUT/spmv_csr_pmc_write.csv > $OUT/spmv_csr_pmc_summary.txt 2>&1
# (b) column-blocked CSR: one launch against one launch per block
cd $REPO
python tools/cb_single_ab.py 1000000 2000000 10000000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostn\|^Librc" > $OUT/column_blocks.txt
# (c) rotation kernels A/B on the headline
for rot in fma mfma; do
  echo "KS_ROTATE=$rot" >> $OUT/rotation.txt
  KS_ROTATE=$rot python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  iterations/s', round(d['value'],1), ' rotate', d['roofline']['per_class']['rotate'])" >> $OUT/rotation.txt
done
ls -la $OUT
# (shift-invert, ks_operator_lu) the two sparse triangular solves of a product on tools/lu_bench.py's 2-D problems
python $REPO/tools/lu_bench.py 200 250 --reps 20 --solve > $OUT/lu_n5e4.txt 2>&1
LU_RESTARTS=6 python $REPO/tools/lu_bench.py 500 1000 --reps 10 --solve > $OUT/lu_n5e5.txt 2>&1
KS_LU_STATS=1 python $REPO/tools/lu_bench.py 500 1000 --reps 1 2>&1 | grep "lu L\|lu U" | tail -2 >> $OUT/lu_n5e5.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_lu -- python $REPO/tools/lu_bench.py 500 1000 --reps 20 > /dev/null 2>> $OUT/bench.err
cp "$(find /tmp/kt_lu -name '*kernel_stats.csv' | head -1)" $OUT/lu_kernel_stats.csv
