#!/bin/bash
# per-kernel averages of a short run (rocprofv3 --kernel-trace --stats): tools/kstats.sh "<command>"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kst && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- $1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/kst/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print(f"{r['Name'][:70]:<70} calls {int(r['Calls']):>6} avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
