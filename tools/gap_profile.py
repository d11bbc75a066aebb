#!/usr/bin/env python3
"""GPU idle time BETWEEN the kernels of an expansion, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python tools/dist_overhead.py 108 plain
    python tools/gap_profile.py /tmp/kt/.../*_kernel_trace.csv
Prints, per kernel class, launches, mean duration, and the mean gap to the PREVIOUS kernel's end."""
import csv
import re
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void ksd::", "", r["Kernel_Name"])
    name = re.sub(r"[<(].*", "", name)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
stat = defaultdict(lambda: [0, 0.0, 0.0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap = (s1 - e0) / 1e3
    if gap > 200:  # restart / host work: not an inter-kernel gap
        continue
    st = stat[n1]
    st[0] += 1
    st[1] += (e1 - s1) / 1e3
    st[2] += max(gap, 0.0)
tot_busy = sum(v[1] for v in stat.values())
tot_gap = sum(v[2] for v in stat.values())
for k, v in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:<22} launches {v[0]:>6}  mean {v[1] / v[0]:8.1f} us  mean gap before {v[2] / v[0]:6.1f} us")
print(f"busy {tot_busy / 1e3:.1f} ms, inter-kernel gaps {tot_gap / 1e3:.1f} ms = {100 * tot_gap / (tot_busy + tot_gap):.1f} % of the expansion time")
