"""Sparse shift-invert on the device (ks_operator_lu): host factorisation (scipy SuperLU), triangular solves on the GPU.
    python tools/lu_bench.py [nx ny [nz]] [--real] [--reps R] [--solve]
Prints fill, dependency levels, ms per product (device) next to the host `lu.solve`, the error against it, and -- with
--solve -- the iterations/s of config-4-style runs (nev 6, 10/20, :LM) with the device operator and the host callback."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import import_package  # noqa: E402

pkg = import_package()


def lap2d(nx, ny):
    ex = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx))
    ey = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(ny, ny))
    return (sp.kron(sp.identity(ny), ex) + sp.kron(ey, sp.identity(nx))).tocsc()


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("dims", nargs="*", type=int, default=[200, 250], help="nx ny [nz]")
    ap.add_argument("--real", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--solve", action="store_true")
    ap.add_argument("--diag", action="store_true")
    ap.add_argument("--unsym", action="store_true", help="add a convection term and factor with SuperLU's defaults (COLAMD, partial pivoting)")
    opt = ap.parse_args()
    nx, ny = opt.dims[0], opt.dims[1]
    nz = opt.dims[2] if len(opt.dims) > 2 else 1
    real, reps = opt.real, opt.reps
    n = nx * ny * nz
    rng = np.random.default_rng(3)
    A = lap2d(nx, ny)
    if nz > 1:  # 3-D: 7-point Laplacian
        ez = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nz, nz))
        A = (sp.kron(sp.identity(nz), A) + sp.kron(ez, sp.identity(nx * ny))).tocsc()
    if real:
        sigma = 1.7
        M = (A - sigma * sp.identity(n)).tocsc()
    else:
        A = (A.astype(np.complex128) + 1j * sp.diags(0.3 * rng.random(n))).tocsc()
        sigma = 1.7 + 0.1j
        M = (A - sigma * sp.identity(n)).tocsc()
    ctx = pkg.Context(0)
    if opt.diag:  # no dependencies at all: what tickets + row start-up cost
        D = sp.diags(1.0 + rng.random(n)).tocsr().astype(M.dtype)
        op = pkg.lu_operator(sp.csr_matrix((n, n), dtype=M.dtype), D, ctx=ctx)
        ws = pkg.ArnoldiWorkspace(n, 4, op.dtype, ctx=ctx)
        ws.set_col(0, np.ones(n, dtype=op.dtype))
        ws.apply(op, 0, 1)
        ctx.synchronize()
        t = time.time()
        for _ in range(reps):
            ws.apply(op, 0, 1)
        ctx.synchronize()
        print(f"n {n} diagonal factors: {1e3 * (time.time() - t) / reps:.3f} ms per product", flush=True)
        return
    if opt.unsym:
        C1 = sp.diags([-0.4, 0.4], [-1, 1], shape=(n, n))
        M = (M + C1.astype(M.dtype)).tocsc()
    t = time.time()
    lu = spla.splu(M) if opt.unsym else spla.splu(M, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    tf = time.time() - t
    t = time.time()
    op = pkg.splu_operator(lu, ctx)
    tu = time.time() - t
    info = op.lu_info
    print(f"n {n} ({nx}x{ny}{'x' + str(nz) if nz > 1 else ''}) {'f64' if real else 'c128'}: factor {tf:.1f} s, upload {tu:.1f} s, {info}", flush=True)
    dt = op.dtype
    ws = pkg.ArnoldiWorkspace(n, 20, dt, ctx=ctx)
    b = (rng.random(n) + (0 if real else 1j * rng.random(n))).astype(dt)
    ws.set_col(0, b)
    ws.apply(op, 0, 1)
    y = ws.col(1)
    t = time.time()
    x = lu.solve(b)
    th = time.time() - t
    print(f"  error vs lu.solve: {np.abs(y - x).max() / np.abs(x).max():.2e}; residual {np.abs(M @ y - b).max():.2e} (host solve: {np.abs(M @ x - b).max():.2e}); host solve {1e3 * th:.1f} ms", flush=True)
    ctx.synchronize()
    t = time.time()
    for _ in range(reps):
        ws.apply(op, 0, 1)
    ctx.synchronize()
    td = (time.time() - t) / reps
    lv = info["levels_l"] + info["levels_u"]
    print(f"  device product {1e3 * td:.3f} ms ({lv} levels: {1e6 * td / lv:.2f} us per level; host/device {th / td:.1f}x)", flush=True)
    y2 = ws.col(1)
    print("  repeatable:", bool(np.array_equal(y, y2)), flush=True)
    if opt.solve:
        kw = dict(nev=6, which="LM", tol=1e-10, mindim=10, maxdim=20, restarts=int(os.environ.get("LU_RESTARTS", "200")))
        t = time.time()
        dec, hist = pkg.partialschur(op, **kw)
        t1 = time.time() - t
        lam = sigma + 1.0 / dec.eigenvalues
        print(f"  device operator: {hist} in {t1:.2f} s = {hist.mvproducts / t1:.1f} iterations/s", flush=True)

        def cb(yv, xv):
            yv[:] = lu.solve(xv)

        hop = pkg.host_operator(cb, n, dt, ctx)
        t = time.time()
        dec2, hist2 = pkg.partialschur(hop, **kw)
        t2 = time.time() - t
        lam2 = sigma + 1.0 / dec2.eigenvalues
        print(f"  host callback:   {hist2} in {t2:.2f} s = {hist2.mvproducts / t2:.1f} iterations/s", flush=True)
        k = min(len(lam), len(lam2))
        if k:
            print("  eigenvalues agree:", float(np.abs(np.sort_complex(lam)[:k] - np.sort_complex(lam2)[:k]).max()), flush=True)


if __name__ == "__main__":
    main()
