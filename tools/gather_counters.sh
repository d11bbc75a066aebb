#!/bin/bash
# L2 request counters of config 3's gather SpMV (k_spmv_csr, column-blocked): how many L2 reads a product issues and how many hit.
#   gpurun --timeout 600 -- 'bash tools/gather_counters.sh'   ->  gpurun_out/gather/  (profiles/r06c_cfg3_gather_counters.txt)
# One rocprofv3 --pmc pass per counter group (only --kernel-trace next to --pmc).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/gather
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $grp | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/g_$tag -- python $REPO/tools/config_bench.py cfg3 --sstep 20 --steps 6 --no-profile > /dev/null 2> $OUT/err_$tag.txt
  f="$(find /tmp/g_$tag -name '*counter_collection.csv' | head -1)"
  python - "$f" "$grp" <<PY
import csv, sys, re
from collections import defaultdict
d = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counter file for", sys.argv[2], e); sys.exit(0)
for r in rows:
    name = re.sub(r"^void ksd::", "", r["Kernel_Name"]); name = re.sub(r"[<(].*", "", name)
    e = d[name][r["Counter_Name"]]; e[0] += 1; e[1] += float(r["Counter_Value"])
for k in ("k_spmv_csr", "k_bupdate_mfma", "k_bdots_mfma"):
    for c, (n, v) in sorted(d.get(k, {}).items()):
        print(f"{k:18s} {c:24s} launches {n:5d}  per launch {v / max(n, 1):14.1f}")
PY
done | tee $OUT/counters.txt
