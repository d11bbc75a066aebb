#!/usr/bin/env python3
"""Restart bubble from a rocprofv3 kernel trace: how long the GPU idles around a Krylov-Schur restart.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python tools/run_solver.py cfg2 25
    python tools/restart_bubble.py /tmp/kt/.../*_kernel_trace.csv [label]

Per restart cycle (identified by the rotation kernel; blit kernels of hipMemcpyAsync are not counted as work):
  bubble A = start(k_rotate*) - end(last expansion kernel before it)   host: sync, fetch H + state, Schur / reorder /
                                                                        restore (src/run.jl:278-360), upload Q
  bubble B = start(first kernel of the next expansion) - end(column copy)   host: bookkeeping, enqueue
Prints medians and the share of the cycle."""
import csv
import re
import statistics
import sys


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = re.sub(r"^void ksd::", "", r["Kernel_Name"])
        name = re.sub(r"[<(].*", "", name)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    return rows


def bubbles(rows):
    A, B, C, last = [], [], [], None
    for i, (s, e, name) in enumerate(rows):
        if not name.startswith("k_rotate"):
            continue
        j = i - 1
        # (blit kernels and the rotation gate -- a one-workgroup kernel that only WAITS for the host, reverse mailbox -- are
        # not work: the bubble runs from the last expansion kernel to the start of the rotation)
        while j >= 0 and (rows[j][2].startswith("__amd") or rows[j][2] == "k_rot_gate"):
            j -= 1
        A.append((s - rows[j][1]) / 1e3)
        j, end = i + 1, e
        while j < len(rows) and (rows[j][2] in ("k_copy", "k_scale") or rows[j][2].startswith("__amd")):
            if not rows[j][2].startswith("__amd"):
                end = rows[j][1]
            j += 1
        if j < len(rows):
            B.append((rows[j][0] - end) / 1e3)
        if last is not None:
            C.append((s - last) / 1e3)
        last = s
    return A[2:-1] or A, B[2:-1] or B, C[2:-1] or C


def main():
    A, B, C = bubbles(load(sys.argv[1]))
    label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    med = statistics.median
    print(f"{label}: {len(A)} restarts | bubble A (expansion end -> rotation start) median {med(A):.1f} us (min {min(A):.1f}) | "
          f"bubble B (restart end -> next expansion) median {med(B):.1f} us | cycle {med(C):.1f} us -> GPU idle {100 * (med(A) + med(B)) / med(C):.1f} % of a cycle")


if __name__ == "__main__":
    main()
