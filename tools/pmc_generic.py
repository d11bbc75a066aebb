#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected SEPARATELY, each with only
--kernel-trace, trimmed to dispatch id / kernel / counter / value by tools/collect_profiles.sh), for workloads whose
launch sequence tools/pmc_summary.py does not model (configs 3 and 4).

    pmc_generic.py fetch.csv write.csv [config_bench.json]

Prints, per kernel (template arguments kept: they say which instantiation ran): launches, average HBM bytes per launch =
2 * FETCH_SIZE + WRITE_SIZE (KiB -> B; the x2 is the gfx950 correction of MI355X_MICROARCH.md for wide coalesced reads;
the SpMV's 4/8-byte gathers are outside that calibration, its figure is an upper-bound-style estimate).  With the
config_bench JSON of the same workload it adds the algorithmic bytes per launch of each kernel class."""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path):
    d = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"^void ksd::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)
        e = d[name]
        e[0] += 1
        e[1] += float(r["Counter_Value"]) * 1024.0
    return d


def klass(name):
    for pre, k in (("k_dots", "dots"), ("k_axpy_dots", "fused"), ("k_axpy", "axpy"), ("k_spmv", "spmv"), ("k_scale", "scale"), ("k_rotate", "rotate")):
        if name.startswith(pre):
            return k
    return None


def main():
    F, W = load(sys.argv[1]), load(sys.argv[2])
    alg = {}
    if len(sys.argv) > 3:
        j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
        for k, v in j["per_class"].items():
            if v.get("GBps"):
                alg[k] = v["GBps"] * 1e9 * v["avg_us"] * 1e-6
    per_class = defaultdict(lambda: [0, 0.0])
    print(f"{'kernel':<58} {'launches':>8} {'fetch x2 MB':>12} {'write MB':>10} {'HBM MB/launch':>14}")
    for name in sorted(F, key=lambda k: -F[k][1]):
        nf, fb = F[name]
        nw, wb = W.get(name, (0, 0.0))
        hbm = (2 * fb / nf) + (wb / nw if nw else 0.0)
        if hbm < 1e5:
            continue
        print(f"{name[:58]:<58} {nf:>8} {2 * fb / nf / 1e6:>12.2f} {(wb / nw if nw else 0) / 1e6:>10.2f} {hbm / 1e6:>14.2f}")
        k = klass(name)
        if k:
            per_class[k][0] += nf
            per_class[k][1] += hbm * nf
    if alg:
        print(f"\n{'class':<8} {'launches':>8} {'HBM MB/launch':>14} {'algorithmic MB/launch':>22} {'ratio':>6}")
        for k, (n, tot) in per_class.items():
            a = alg.get(k)
            print(f"{k:<8} {n:>8} {tot / n / 1e6:>14.2f} {(a or 0) / 1e6:>22.2f} {(tot / n / a if a else 0):>6.2f}")


if __name__ == "__main__":
    main()
