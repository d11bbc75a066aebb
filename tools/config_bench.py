#!/usr/bin/env python3
"""Per-kernel roofline of BASELINE.json's OTHER single-GPU configurations (bench.py is the headline one), same state
machine as bench.py: initial expansion, W warm-up restart cycles, K timed ones, then the same K cycles with HIP events
around every launch (library stream) for the per-class figures.

    python tools/config_bench.py CONFIG [--steps K] [--warmup W]

CONFIG
  cfg2    3-D 7-point Laplacian 100^3 (n = 10^6), nev = 20, :SR, mindim/maxdim 20/40, Float64
  cfg3    hashed nonsymmetric n = 10^6, ~5 entries per row, nev = 10, :LM, mindim/maxdim 10/20, Float64 (complex Ritz pairs)
  cfg3x   the same at n = 10^7
  cfg4    ComplexF64, n = 5*10^5, nev = 6, mindim/maxdim 10/20, :LM on a DEVICE-resident complex band operator: the
          device side of config 4 (all expansion kernels in ComplexF64) without the PCIe round trip of the host LU
  cfg4big the same at n = 5*10^6 (column = 80 MB: the ComplexF64 kernels outside the launch-bound regime)
  cfg4d   config 4 with the shift-invert operator ON THE DEVICE: (A - sigma I)^{-1} x by rocSPARSE's pivoting tridiagonal
          solver through the device-callback operator (arnoldimethod.jl_amd/extras.py); nothing n-sized crosses PCIe
  cfg4h   config 4 proper: shift-invert through an opaque HOST operator (scipy splu of the shifted tridiagonal matrix),
          every product staged over PCIe (docs/src/index.md:246-249)
Prints ONE JSON line: iterations/s, per-class {launches, avg us, GB/s, frac of 8 TB/s}, moved-bytes figure of the
expansion, and the SpMV layout.  Run under `rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` for profiles/."""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
M = ks.matrices
PEAK = 8000.0


def build(config, ctx):
    if config == "cfg2":
        n = 100 ** 3
        A = M.to_scipy(*M.laplace3d_csr(100, 100, 100), n)
        return ks.csr_operator(A, ctx), n, A.nnz, np.float64, dict(nev=20, which="SR", mindim=20, maxdim=40), "laplace3d 100^3"
    if config in ("cfg3", "cfg3x"):
        n = 1_000_000 if config == "cfg3" else 10_000_000
        A = M.hashed_nonsymmetric_csr(n, seed=7)
        return ks.csr_operator(A, ctx), n, A.nnz, np.float64, dict(nev=10, which="LM", mindim=10, maxdim=20), f"hashed nonsymmetric n={n}"
    n = 5_000_000 if config == "cfg4big" else 500_000
    rng = np.random.default_rng(0)
    A = (M.to_scipy(*M.laplace1d_csr(n), n) + 1j * sp.diags(0.3 * rng.random(n))).tocsr().astype(np.complex128)
    prm = dict(nev=6, which="LM", mindim=10, maxdim=20)
    if config in ("cfg4", "cfg4big"):
        return ks.csr_operator(A, ctx), n, A.nnz, np.complex128, prm, f"complex tridiagonal n={n} (device-resident operator)"
    sigma = 1.7 + 0.1j
    if config == "cfg4d":
        import torch  # noqa: F401 - must be loaded before the library touches the device (two HIP runtimes in one process)
        from arnoldimethod_jl_amd import extras

        T = A.tocsr()
        si = extras.TridiagonalShiftInvert(T.diagonal(-1), T.diagonal(0), T.diagonal(1), sigma, ctx)
        return si.operator, n, 0, np.complex128, prm, "shift-invert ON THE DEVICE (rocSPARSE zgtsv through the device-callback operator) n=5e5"
    import scipy.sparse.linalg as spla

    lu = spla.splu((A - sigma * sp.identity(n)).tocsc())

    def mul(y, x):
        y[:] = lu.solve(x)

    return ks.host_operator(mul, n, np.complex128, ctx), n, 0, np.complex128, prm, "shift-invert (host LU callback) n=5e5"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-profile", action="store_true", help="skip the second (HIP-event instrumented) pass: under a kernel trace the timed cycles are then the last ones")
    ap.add_argument("--sync-cycles", action="store_true", help="synchronise the device after every timed cycle (as all records before round 6b did); "
                    "default: cycles back to back like the library's own driver, the timed region synchronised at both ends")
    ap.add_argument("--sstep", type=int, default=int(os.environ.get("KS_BENCH_SSTEP", "0")),
                    help="s-step (block) expansion: steps per block (ks_workspace_set_sstep); 0 = per-step expansion")
    args = ap.parse_args()
    if args.config == "cfg4d":
        import torch  # noqa: F401 - before the first HIP call of the library
    ctx = ks.Context(0)
    op, n, nnz, dtype, prm, what = build(args.config, ctx)
    esz = np.dtype(dtype).itemsize
    nev, which, mindim, maxdim = prm["nev"], prm["which"], prm["mindim"], prm["maxdim"]
    tol = float(np.sqrt(np.finfo(np.float64).eps))
    ws = ks.ArnoldiWorkspace(n, maxdim, dtype, ctx=ctx)
    v1 = M.start_vector(n).astype(dtype)
    if esz == 16:
        v1 = v1 + 1j * M.start_vector(n, seed=5)
    ws.set_sstep(args.sstep if args.sstep >= 2 else 0)   # (0 included: the library's own default is ON)
    ws.reinitialize(0, v1)
    ws.iterate_arnoldi(op, 1, mindim)
    fmt = op.format
    state = dict(k=mindim, active=0, steps=0, moved=0.0, t_expand=0.0, t_restart=0.0, reorth=0)
    spmv_b = fmt["bytes_per_nnz"] * nnz + 4.0 * (n + 1) + 2.0 * esz * n if nnz else 0.0

    def cycle(timed, sync=True):
        k = state["k"]
        t0 = time.perf_counter()
        if os.environ.get("KS_BENCH_SPLIT_CYCLE", "0") == "1":  # the two calls of rounds 1-2
            st = ws.iterate_arnoldi(op, k + 1, maxdim)
            t1 = time.perf_counter()
            r = ws.restart(state["active"], nev, which, tol, mindim, maxdim)
        else:  # one cycle the way ks_partialschur runs it (early part of the host step overlapped with the expansion's tail)
            r = st = ws.expand_restart(op, k, state["active"], nev, which, tol, mindim, maxdim)
            t1 = t0 + r["seconds"][0]
        if sync:
            ctx.synchronize()
        t2 = time.perf_counter()
        if timed:
            state["steps"] += maxdim - k
            info = ws.sstep_info
            blk = ks.sstep_partition(dtype, k + 1, maxdim - k, info["s"]) if (args.sstep >= 2 and info["s"] >= 2) else []
            if blk and info["blocks"] - state.get("blocks_seen", 0) != len(blk):
                blk = []                                   # (a block was abandoned, or no shifts yet: not a block cycle)
            state["blocks_seen"] = info["blocks"]
            state["abandoned"] = info["abandoned"]
            state["blk_cycles"] = state.get("blk_cycles", 0) + (1 if blk else 0)
            if not blk:
                state["reorth"] += st["reorth"]            # (DGKS second passes exist on per-step cycles only)
            if blk:  # per block of s steps on kk columns: s products (+ 3 esz n for the shift unless fused) + two passes
                shift_b = 0.0 if fmt["layout"] == "stencil" else 3.0 * esz * n
                kk = k + 1
                for sb in blk:
                    state["moved"] += sb * (spmv_b + shift_b) + esz * n * (kk + sb) + esz * n * (kk + 2 * sb)
                    kk += sb
            for j in range(k + 1, maxdim + 1):
                if not blk:
                    state["moved"] += spmv_b + esz * n * (j + 1) + esz * n * (j + 2)
            # (a third pass over V exists only on the explicit-second-pass path, KS_PASSES=3: the default expansion carries
            # the second projection in the triangular factor -- booking it unconditionally gave moved_frac 1.07 in round 3)
            if ws.passes == 3 and not blk:
                state["moved"] += st["reorth"] * esz * n * ((k + 1 + maxdim) / 2.0 + 2)
            state["t_expand"] += t1 - t0
            state["t_restart"] += t2 - t1
        state["k"], state["active"] = r["k"], min(r["nlock"], nev - 1)  # keep cycling even if everything converged

    for _ in range(args.warmup):
        cycle(False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cycle(True, sync=args.sync_cycles)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    if not args.sync_cycles:   # (the tail the closing synchronisation waited for belongs to the timed region: bench.py does the same)
        state["t_restart"] += max(0.0, elapsed - (state["t_expand"] + state["t_restart"]))
    # the reference's two invariants on the benched workspace (test/expansion.jl:29-30), as bench.py's `validation`
    rel, orth = ws.arnoldi_relation(op, state["k"]) if nnz else (float("nan"), float("nan"))
    hnorm = float(np.linalg.norm(np.array(ws.H)[: state["k"] + 1, : state["k"]]))
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(0 if args.no_profile else args.steps):
        cycle(False)
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    per = {}
    # the second-pass update (class "axpy") skips itself on the device when the DGKS test did not ask for it
    # (src/expansion.jl:91): its bytes were booked at enqueue time -> scale by the fraction of steps that ran it
    if prof["axpy"]["count"]:
        prof["axpy"]["bytes"] *= min(1.0, state["reorth"] / max(1, state["steps"]))
    for k_, v in prof.items():
        if v["count"]:
            gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 and v["bytes"] > 0 else None
            per[k_] = {"launches": v["count"], "avg_us": 1e3 * v["ms"] / v["count"], "GBps": gbs, "frac": gbs / PEAK if gbs else None}
    # bytes over the WHOLE cycle time: speculative products of the next expansion run during the restart interval (bench.py,
    # traffic_fractions), so the expansion interval alone is not what the bytes of a cycle were moved in
    moved = state["moved"] / max(state["t_expand"] + state["t_restart"], 1e-12) / 1e9
    out = {
        "config": args.config, "workload": f"{what}, nev={nev}, which={which}, mindim={mindim}, maxdim={maxdim}, dtype={'c128' if esz == 16 else 'f64'}",
        "iters_per_s": state["steps"] / elapsed, "ms_per_cycle": 1e3 * elapsed / args.steps, "iterations": state["steps"],
        "dgks_second_passes": state["reorth"], "cycles_synchronised_one_by_one": bool(args.sync_cycles), "spmv_layout": fmt, "per_class": per,
        "sstep": {"requested": args.sstep, "in_force": ws.sstep_info["s"], "block_cycles": state.get("blk_cycles", 0), "abandoned": state.get("abandoned", 0),
                  **{k: ws.sstep_info[k] for k in ("fused_rotations", "split_rotations", "chains_adopted", "chains_dropped", "gram_dev")}} if args.sstep >= 2 else None,
        "validation": {"arnoldi_rel": rel / hnorm if hnorm else None, "orth": orth, "k": state["k"], "locked": state["active"]},
        "expansion": {"moved_GBps": moved, "moved_frac": moved / PEAK, "expand_seconds": state["t_expand"], "restart_seconds": state["t_restart"]},
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
