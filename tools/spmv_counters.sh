#!/bin/bash
# Cache counters of the stencil product: tools/spmv_cold_warm.py under rocprofv3 --pmc, one counter set per
# pass (with --kernel-trace only).  Prints per phase of the probe (warm / flushed / produced / chain; 20 launches each, in that
# order after the 3 first ones) the average counter values per launch of k_spmv_stencil2.
#   gpurun --timeout 900 -- 'bash tools/spmv_counters.sh > gpurun_out/spmv_counters_r04.txt 2>&1'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $set | tr ' ' '_')
  rm -rf /tmp/spc_$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/spc_$tag -- python $REPO/tools/spmv_cold_warm.py > /tmp/spc_$tag.log 2>&1
  f=$(find /tmp/spc_$tag -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "## $set: no counter file (counter not available on this device?)"; tail -3 /tmp/spc_$tag.log; continue; fi
  python - "$f" "$set" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_spmv_stencil2" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
names = sys.argv[2].split()
print("##", sys.argv[2], "-", len(ids), "launches of k_spmv_stencil2")
phases = [("first", 0, 3), ("warm", 3, 23), ("flushed", 23, 43), ("produced", 43, 63), ("chain", 63, 83)]
for name, a, b in phases:
    sel = ids[a:b]
    if not sel:
        continue
    print(f"  {name:9s}", "  ".join(f"{n} {sum(by[i].get(n, 0.0) for i in sel) / len(sel):14.0f}" for n in names))
PY
done
