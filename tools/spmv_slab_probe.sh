#!/bin/bash
# probes of the slab SpMV on one MI355X: which part of a step costs the time (dbg bits: 1 no arithmetic, 2 no stores, 4 no x copies,
# 8 no mask copies), and PMC counters of the library kernel (variant 0) against slab variants.  gpurun -- 'bash tools/spmv_slab_probe.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BIN=$REPO/tools/_build/spmv_slab
for dbg in 0 1 2 3 4 6 7 12 15; do
  echo "## dbg=$dbg" >> $OUT/probe_dbg.txt
  timeout 120 $BIN 216 216 216 0 0 $dbg 2>&1 | grep "shifted=1" | grep -v DIFFERx >> $OUT/probe_dbg.txt
done
for var in 0 1 5 7 10; do
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
    d=/tmp/pmc_${var}_$(echo $ctr | tr ' ' '_' | cut -c1-40)
    rm -rf $d
    timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- $BIN 216 216 216 only $var 12 > /dev/null 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    echo "## variant $var: $ctr" >> $OUT/probe_pmc.txt
    if [ -n "$f" ]; then
      python3 - "$f" >> $OUT/probe_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "flush" in k: continue
    a = acc[(k[:60], row["Counter_Name"])]
    a[0] += float(row["Counter_Value"]); a[1] += 1
for (k, c), (v, cnt) in sorted(acc.items()):
    print(f"  {k:60s} {c:32s} per launch {v / cnt:16.1f}  ({cnt} launches)")
PY
    else echo "  (no counter file)" >> $OUT/probe_pmc.txt; fi
  done
done
cat $OUT/probe_dbg.txt
cat $OUT/probe_pmc.txt
