#!/usr/bin/env python3
"""The kernels of a few restart cycles in launch order, from a rocprofv3 kernel trace: start (us from the window's first kernel),
duration, idle gap to the previous kernel's end.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python tools/config_bench.py cfg3 --sstep 20 --no-profile
    python tools/kernel_timeline.py /tmp/kt/.../*_kernel_trace.csv [first_fraction=0.5] [count=120]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void ksd::", "", r["Kernel_Name"])
    name = re.sub(r"[<(].*", "", name)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 120
i0 = int(len(rows) * frac)
base = rows[i0][0]
prev_end = rows[i0 - 1][1] if i0 else rows[0][0]
busy = idle = 0.0
for s, e, n in rows[i0:i0 + cnt]:
    gap = (s - prev_end) / 1e3
    print(f"{(s - base) / 1e3:10.1f} us  {n:<28} {(e - s) / 1e3:8.1f} us   gap {gap:7.1f}")
    busy += (e - s) / 1e3
    idle += max(gap, 0.0)
    prev_end = e
print(f"window: busy {busy:.0f} us, idle {idle:.0f} us")
