#!/usr/bin/env python3
"""SpMV micro-benchmark: y = A x through the product's operator (ks_apply), timed with the library's own HIP events
(class "spmv"), for the matrices of BASELINE.json's configurations in every device layout.

    python tools/spmv_bench.py [case ...]        cases: lap216 lap100 lapvar216 hashed1e6 hashed1e7 skew1e6 cplx5e5 (default: all but hashed1e7)

Prints one line per (matrix, layout): microseconds per application, GB/s on the bytes THAT layout streams
(bytes_per_nnz * nnz + 4 (n+1) + 2 * sizeof(T) * n), the fraction of the 8 TB/s HBM3E spec, and the same launch
priced at plain CSR's 12 B / non-zero (SURVEY 8d's formula).  Used for profiles/r02_spmv_*.txt; run it under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` to get the traffic next to it."""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
M = ks.matrices
REPS = int(os.environ.get("SPMV_REPS", "40"))
TRIALS = int(os.environ.get("SPMV_TRIALS", "3"))


def skewed(n, seed=3):
    """cfg-3-like matrix with a heavy tail: the hashed nonsymmetric matrix plus 64 rows of 40 000 entries and a band of
    2 000 rows with 300 entries each (same total order of non-zeros as the balanced one + ~3.2e6)."""
    A = M.hashed_nonsymmetric_csr(n, seed=7)
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for r in rng.choice(n, 64, replace=False):
        c = rng.choice(n, 40_000, replace=False)
        rows.append(np.full(c.shape, r))
        cols.append(c)
    for r in range(n // 2, n // 2 + 2000):
        c = rng.choice(n, 300, replace=False)
        rows.append(np.full(c.shape, r))
        cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    B = sp.coo_matrix((rng.random(rows.shape[0]), (rows, cols)), shape=(n, n)).tocsr()
    C = (A + B).tocsr()
    C.sort_indices()
    return C


def cases():
    yield "lap216", lambda: M.to_scipy(*M.laplace3d_csr(216, 216, 216), 216 ** 3), ("csr", "vi", "sell", "sellvi", "dvi", "stencil")
    yield "lap100", lambda: M.to_scipy(*M.laplace3d_csr(100, 100, 100), 100 ** 3), ("csr", "vi", "sell", "sellvi", "dvi", "stencil")
    def lapvar():
        """7-point stencil with VARIABLE coefficients (more than 256 distinct values: no dictionary layout applies) --
        what the default selection turns into sliced ELLPACK"""
        ip, ix, dv = M.laplace3d_csr(216, 216, 216)
        dv = dv * (1.0 + 0.5 * M.uniform_hash(3, np.arange(dv.shape[0])))
        return M.to_scipy(ip, ix, dv, 216 ** 3)

    yield "lapvar216", lapvar, ("", "csr")
    yield "hashed1e6", lambda: M.hashed_nonsymmetric_csr(1_000_000, seed=7), ("csr", "sell")
    yield "hashed2e5", lambda: M.hashed_nonsymmetric_csr(200_000, seed=7), ("csr",)
    yield "hashed5e5", lambda: M.hashed_nonsymmetric_csr(500_000, seed=7), ("csr",)
    yield "hashed2e6", lambda: M.hashed_nonsymmetric_csr(2_000_000, seed=7), ("csr",)
    yield "skew1e6", lambda: skewed(1_000_000), ("csr",)
    yield "cplx5e5", lambda: (M.to_scipy(*M.laplace1d_csr(500_000), 500_000) + 1j * sp.diags(0.3 * np.cos(np.arange(500_000)))).tocsr().astype(np.complex128), ("csr",)
    yield "hashed1e7", lambda: M.hashed_nonsymmetric_csr(10_000_000, seed=7), ("csr",)


def main():
    want = sys.argv[1:] or ["lap216", "lap100", "hashed1e6", "skew1e6", "cplx5e5"]
    ctx = ks.Context(0)
    print(f"{'matrix':<10} {'layout':<8} {'n':>9} {'nnz':>10} {'max row':>8} {'us':>8} {'GB/s':>8} {'frac':>6} {'csr-equivalent GB/s':>20}  env={ {k: v for k, v in os.environ.items() if k.startswith('KS_')} }")
    for name, build, layouts in cases():
        if name not in want:
            continue
        A = build()
        n, nnz = A.shape[0], A.nnz
        dt = np.complex128 if A.dtype.kind == "c" else np.float64
        esz = 16 if dt == np.complex128 else 8
        maxrow = int(np.diff(A.indptr).max())
        ws = ks.ArnoldiWorkspace(n, 2, dt, ctx=ctx)
        ws.set_col(0, (M.start_vector(n) + (1j * M.start_vector(n, seed=5) if esz == 16 else 0)).astype(dt))
        ref = A @ ws.col(0)
        for lay in layouts:
            os.environ.pop("KS_SPMV_FORMAT", None)
            if lay:
                os.environ["KS_SPMV_FORMAT"] = lay
            # the speed of a streaming kernel on this part depends on WHICH physical pages back its buffers (+-6 %,
            # DESIGN.md section 3): time TRIALS uploads of the same matrix with a dummy allocation in between, report the
            # fastest and the spread
            times, hold = [], []
            for trial in range(TRIALS):
                op = ks.csr_operator(A, ctx)
                fmt = op.format
                got = fmt["layout"]
                for _ in range(3):
                    ws.apply(op, 0, 1)
                if trial == 0:
                    err = np.abs(ws.col(1) - ref).max() / max(1e-300, np.abs(ref).max())
                ctx.profile_reset()
                ctx.profile_enable(True)
                for _ in range(REPS):
                    ws.apply(op, 0, 1)
                p = ctx.profile_get()["spmv"]
                ctx.profile_enable(False)
                times.append(1e3 * p["ms"] / p["count"])
                op.close()
                if TRIALS > 1:  # ~100, 200, ... MB kept allocated so that the next upload lands elsewhere
                    hold.append(ks.ArnoldiWorkspace((trial + 1) * 6_100_000, 1, np.float64, ctx=ctx))
            for h in hold:
                h.close()
            us = min(times)
            gbs = p["bytes"] / p["count"] / (us * 1e-6) / 1e9
            eq = ((4 + esz) * nnz + 4 * (n + 1) + 2 * esz * n) / (us * 1e-6) / 1e9
            print(f"{name:<10} {got:<8} {n:>9} {nnz:>10} {maxrow:>8} {us:>8.1f} {gbs:>8.0f} {gbs / 8000:>6.3f} {eq:>20.0f}  relerr={err:.1e}  "
                  f"trials(us)={' '.join('%.1f' % t for t in times)}", flush=True)
        os.environ.pop("KS_SPMV_FORMAT", None)
        ws.close()


if __name__ == "__main__":
    main()
