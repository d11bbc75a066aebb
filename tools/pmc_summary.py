#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel and compare with the algorithmic
bytes of each launch (n, nnz, j recovered from the kernel name / launch order).

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <n> <nnz>

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE is in KiB and reports exactly 1/2 of
the bytes of a wide (16 B/lane) coalesced streaming read -> x2 for such kernels; WRITE_SIZE (KiB) is
uncalibrated by the guide -- calibrated here on k_copy / k_scale whose written bytes are known."""
import csv
import re
import sys
from collections import defaultdict


def load(path):
    per = defaultdict(list)
    order = []
    for r in csv.DictReader(open(path)):
        name = re.sub(r"void ksd::|\(.*", "", r["Kernel_Name"])
        per[name].append(float(r["Counter_Value"]))
        order.append((int(r["Dispatch_Id"]), name, float(r["Counter_Value"])))
    return per, sorted(order)


def main():
    fpath, wpath, n, nnz = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    F, fo = load(fpath)
    W, wo = load(wpath)
    # recover j per dots/axpy launch from launch order: each step = spmv, dots(j), fin, fused(j), fin, fin, axpy(j), fin, scale
    print(f"{'kernel':30s} {'launches':>8s} {'FETCH KiB avg':>14s} {'x2 -> GB':>10s} {'WRITE KiB avg':>14s} {'GB':>8s}")
    for name in sorted(F, key=lambda k: -sum(F[k])):
        f = sum(F[name]) / len(F[name])
        w = sum(W.get(name, [0])) / max(len(W.get(name, [0])), 1)
        print(f"{name:30s} {len(F[name]):8d} {f:14.1f} {2*f*1024/1e9:10.3f} {w:14.1f} {w*1024/1e9:8.3f}")
    col = 8.0 * n
    print("\nreference byte counts: one column = %.4f GB; matrix stream (12 nnz + 4 (n+1)) = %.4f GB" % (col / 1e9, (12.0 * nnz + 4 * (n + 1)) / 1e9))


if __name__ == "__main__":
    main()
