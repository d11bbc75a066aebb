#!/usr/bin/env python3
"""Runs the five BASELINE.json configurations (4 of them on one GPU) end to end through the public API and
prints convergence history, wall time and on-device residuals.  Not a bench contract, a health check."""
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


def report(tag, A_op, dec, hist, t):
    res, orth = dec.workspace.residual_norms(A_op, dec.nconverged) if dec.nconverged else (0.0, 0.0)
    print(f"[{tag}] {hist} | restarts {hist.restarts} reorth {hist.reorth} | {t:.3f} s "
          f"(expand {hist.seconds_expand:.3f} host {hist.seconds_host:.3f} rotate {hist.seconds_rotate:.3f}) | "
          f"||AQ-QR||={res:.2e} ||Q'Q-I||={orth:.2e} | iters/s {hist.mvproducts / t:.0f}", flush=True)


def cfg1():
    A = M.to_scipy(*M.laplace1d_csr(100), 100)
    t = time.perf_counter()
    dec, hist = ks.partialschur(A, nev=10, which="SR", tol=1e-6, v1=M.start_vector(100))
    report("cfg1 tridiag-100 SR nev10", ks.as_operator(A), dec, hist, time.perf_counter() - t)


def cfg2(m=100, restarts=30, tol=None):
    n = m ** 3
    A = M.to_scipy(*M.laplace3d_csr(m, m, m), n)
    op = ks.csr_operator(A)
    ws = ks.ArnoldiWorkspace(M.start_vector(n), 40)
    t = time.perf_counter()
    dec, hist = ks.partialschur_(op, ws, nev=20, which="SR", restarts=restarts, tol=tol)
    report(f"cfg2 laplace3d-{m}^3 SR nev20 ({restarts} restarts)", op, dec, hist, time.perf_counter() - t)


def cfg3(n=1_000_000):
    A = M.hashed_nonsymmetric_csr(n, seed=7)
    op = ks.csr_operator(A)
    ws = ks.ArnoldiWorkspace(M.start_vector(n), 20)
    t = time.perf_counter()
    dec, hist = ks.partialschur_(op, ws, nev=10, which="LM", restarts=60)
    report(f"cfg3 hashed-nonsym n={n} LM nev10", op, dec, hist, time.perf_counter() - t)
    print("      eigenvalues:", np.array2string(dec.eigenvalues[:4], precision=6))


def cfg4(n=500_000):
    import scipy.sparse.linalg as spla

    rng = np.random.default_rng(0)
    A = (M.to_scipy(*M.laplace1d_csr(n), n) + 1j * sp.diags(0.3 * rng.random(n))).tocsc().astype(np.complex128)
    sigma = 1.7 + 0.1j
    t0 = time.perf_counter()
    lu = spla.splu((A - sigma * sp.identity(n)).tocsc())
    tf = time.perf_counter() - t0

    class ShiftInvert:
        shape = (n, n)
        dtype = np.complex128

        def mul_(self, y, x):
            y[:] = lu.solve(x)

    v1 = M.uniform_hash(1, np.arange(n)) + 1j * M.uniform_hash(2, np.arange(n))
    t = time.perf_counter()
    dec, hist = ks.partialschur(ShiftInvert(), v1=v1, nev=6, which="LM", tol=1e-10)
    dt = time.perf_counter() - t
    lam = sigma + 1.0 / dec.eigenvalues
    Q = dec.Q
    r = max(np.linalg.norm(A @ Q[:, i] - lam[i] * Q[:, i]) for i in range(1)) if dec.nconverged else float("nan")
    print(f"[cfg4 complex shift-invert n={n} nev6 LM via host callback] {hist} | {dt:.3f} s (LU {tf:.2f} s) | "
          f"lambda[0]={lam[0]:.8f} resid(A q0 - lam q0)={r:.2e}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["1", "2", "3", "4"]
    if "1" in which:
        cfg1()
    if "2" in which:
        cfg2()
    if "2b" in which:
        cfg2(216, restarts=10)
    if "3" in which:
        cfg3()
    if "4" in which:
        cfg4()
