#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected in SEPARATE runs, each with only
--kernel-trace next to --pmc) per kernel, and write profiles/pmc_traffic.json for bench.py.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <n> <nnz> [out.json] [bytes_per_nnz] [index_bytes_per_row] [rotation_columns] [kstart]
(out.json is MERGED: classes of an earlier pass that this one does not exercise are kept)
(bytes_per_nnz: what the SpMV layout in use streams per stored entry -- 12 plain CSR, 4 value-indexed, 1
delta-value-indexed; `bench.py` prints it as config.spmv_layout)

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE is in KiB and reports exactly 1/2 of the
bytes of a wide (16 B/lane) coalesced streaming read -> x2 for the streaming kernels (k_dots, k_axpy*,
k_scale, k_rotate, k_copy; verified here: k_scale/k_copy/k_norm2 read exactly one column = 8n bytes and
report 4n).  The same factor is applied to k_spmv_csr, whose 4- and 8-byte loads are NOT covered by the
guide's calibration -- its figure is flagged "uncalibrated".  WRITE_SIZE (KiB) needs no correction: k_copy,
k_scale and the SpMV write exactly 8n bytes and report 8n.
Algorithmic bytes per launch are recomputed from the launch order (the basis size j of each step)."""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = re.sub(r"void ksd::|\(.*", "", r["Kernel_Name"])
        rows.append((int(r["Dispatch_Id"]), name, float(r["Counter_Value"])))
    rows.sort()
    return rows


def klass(name):
    if name.startswith("k_brotdots"):
        return "blk_rotate"
    if name.startswith("k_bdots"):
        return "blk_dots"
    if name.startswith("k_bupdate"):
        return "blk_fused"
    if name.startswith("k_dots"):
        return "dots"
    if name.startswith("k_axpy_dots"):
        return "fused"
    if name.startswith("k_axpy"):
        return "axpy"
    if name.startswith("k_spmv"):
        return "spmv"
    if name.startswith("k_scale"):
        return "scale"
    if name.startswith("k_rotate"):
        return "rotate"
    return None


def main():
    fpath, wpath, n, nnz = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    out = sys.argv[5] if len(sys.argv) > 5 else None
    bpn = float(sys.argv[6]) if len(sys.argv) > 6 else 12.0
    aux = float(sys.argv[7]) if len(sys.argv) > 7 else 4.0  # index bytes per row next to the non-zeros (4: rowptr; 0: stencil-mask layout)
    rot_cols = float(sys.argv[8]) if len(sys.argv) > 8 else 60.0  # columns the restart rotation reads + writes (T-folded: 41 + 21)
    kstart = int(sys.argv[9]) if len(sys.argv) > 9 else 21       # s-step expansion: columns in place when a batch starts (k + 1 of the restart)
    def after_calibration(rows):
        # the launches before the first SpMV are the placement search of ks_workspace_create; their number
        # differs from process to process
        for i, r in enumerate(rows):
            if klass(r[1]) == "spmv":
                return rows[i:]
        return rows

    F, W = after_calibration(load(fpath)), after_calibration(load(wpath))
    assert [x[1] for x in F] == [x[1] for x in W], "the two passes must replay the same launch sequence"
    col = 8.0 * n
    # basis size j of a step = number of k_fin_dots/k_fin_mid workgroups is not in the trace; recover it from
    # the dots template parameter and the order inside one expansion (j increases by one per step).
    per = defaultdict(lambda: dict(launches=0, fetch=0.0, write=0.0, alg=0.0))
    j = None
    blk_k = None
    last_nc4 = None
    started = False  # the launches before the first SpMV are the placement calibration of ks_workspace_create
    for (_, name, f), (_, _, w) in zip(F, W):
        k = klass(name)
        if k is None:
            continue
        if k == "spmv":
            started = True
        if not started:
            continue
        if k == "spmv":
            alg = bpn * nnz + aux * (n + 1) + 2 * col
        elif k == "blk_rotate":
            # restart rotation fused with the first pass (k_brotdots_mfma<NGX, NTK, NT>): reads the old basis and the block,
            # writes the rotated columns; the block that follows starts on `kstart` columns
            S = 4 * int(re.search(r"k_brotdots_mfma<\d+, \d+, (\d+)[,>]", name).group(1))   # (<NGX, NTK, NT[, CX]>)
            blk_k = kstart
            alg = col * (rot_cols + S)
        elif k in ("blk_dots", "blk_fused"):
            # s-step kernels: k_bdots<double, NCW, S, ...> reads the kb existing columns and the S new ones, k_bupdate reads the
            # same and writes the S; kb starts at `kstart` after every rotation and grows by S per block
            if re.search(r"k_b(?:dots|update)_mfma<", name):
                S = 4 * int(re.search(r"_mfma<\d+, (\d+)[,>]", name).group(1))                       # (matrix-instruction forms: <NGS, NT>; s = 4 NT on the headline)
            elif re.search(r"k_b(?:dots|update)_ringL<", name):
                S = 20                                                                            # (large-block ring forms: <NCW>)
            else:
                mm = re.search(r"k_b(?:dots|update)(?:_ring<\d+, (\d+)|<double, \d+, (\d+))", name)   # (ring forms: <NCW, S, NW, WB>)
                S = int(mm.group(1) or mm.group(2))
            if k == "blk_dots":
                kb = kstart if (blk_k is None) else blk_k
                blk_k = kb
                alg = col * (kb + S)
            else:
                alg = col * (blk_k + 2 * S)
                blk_k += S
        elif k == "dots":
            nc4 = int(re.search(r"k_dots<double, (\d+)", name).group(1))
            # first step of an expansion: smallest j of the granule is unknown -> track by sequence
            if j is None or nc4 < (last_nc4 or 0):
                j = None
            last_nc4 = nc4
            j = (j + 1) if j is not None and (j + 1 + 3) // 4 == nc4 else (4 * (nc4 - 1) + 1 if j is None else j + 1)
            alg = col * (j + 1)
        elif k == "fused":
            alg = col * (j + 2)
        elif k == "axpy":
            alg = col * (j + 2)
        elif k == "scale":
            alg = 2 * col
        elif k == "rotate":
            blk_k = None
            alg = col * rot_cols  # the restart of the bench workload reads 40 columns and writes 20 (T-folded form: 41 and 21)
        else:
            alg = None
        e = per[k]
        e["launches"] += 1
        e["fetch"] += f * 1024.0
        e["write"] += w * 1024.0
        if alg is not None:
            e["alg"] += alg
    res = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes; FETCH x2 (gfx950)", "n": n, "nnz": nnz, "classes": {}}
    print(f"{'class':8s} {'launches':>8s} {'HBM GB/launch (2F+W)':>22s} {'algorithmic GB/launch':>22s} {'ratio':>6s}")
    for k, e in per.items():
        hbm = (2 * e["fetch"] + e["write"]) / e["launches"]
        alg = e["alg"] / e["launches"] if e["alg"] else None
        res["classes"][k] = {"launches": e["launches"], "hbm_bytes_per_launch": hbm, "fetch_bytes_raw_per_launch": e["fetch"] / e["launches"],
                             "write_bytes_per_launch": e["write"] / e["launches"], "algorithmic_bytes_per_launch": alg,
                             "calibrated": k != "spmv"}
        print(f"{k:8s} {e['launches']:8d} {hbm/1e9:22.4f} {(alg or 0)/1e9:22.4f} {(hbm/alg if alg else 0):6.3f}")
    if out:
        try:  # keep the classes of an earlier pass that this one did not exercise (per-step kernels next to the block kernels)
            old = json.load(open(out))
            for kk, vv in old.get("classes", {}).items():
                res["classes"].setdefault(kk, vv)
        except Exception:
            pass
        json.dump(res, open(out, "w"), indent=1)
        print("wrote", out)


if __name__ == "__main__":
    main()
