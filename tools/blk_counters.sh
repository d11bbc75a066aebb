#!/bin/bash
# SQ / LDS counters of the ring kernels of the s-step expansion (k_bdots_ring, k_bupdate_ring) on the 216^3 basis:
# tools/blk_bench.py (s = 10 shapes) under rocprofv3 --pmc, one counter set per pass.  Average per launch.
#   gpurun --timeout 900 -- 'bash tools/blk_counters.sh > gpurun_out/blk_counters_r04.txt 2>&1'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export BLK_S=${BLK_S:-10} BLK_DBGS=0 BLK_DBGS0=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_')
  rm -rf /tmp/bkc_$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/bkc_$tag -- python $REPO/tools/blk_bench.py > /tmp/bkc_$tag.log 2>&1
  f=$(find /tmp/bkc_$tag -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "## $set: no counter file"; tail -2 /tmp/bkc_$tag.log; continue; fi
  python - "$f" "$set" <<'PY'
import csv, sys, collections, re
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"k_b(dots|update)_ring(L?)<(\d+)", r["Kernel_Name"])
    if m:
        by[f"k_b{m.group(1)}_ring{m.group(2)}<NCW={m.group(3)}>"][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("##", sys.argv[2])
for k in sorted(by):
    print(f"  {k:28s}", "  ".join(f"{n} {sum(v) / len(v):14.0f}" for n, v in sorted(by[k].items())), f"({len(next(iter(by[k].values())))} launches)")
PY
done
