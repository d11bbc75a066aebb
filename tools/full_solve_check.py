#!/usr/bin/env python3
"""A whole solve to convergence at full size through the library's driver, blocks on and off: products, restarts, wall time,
device-side residual ||A Q - Q R||_F and orthogonality of the converged vectors, Ritz values against the analytic spectrum.
    python tools/full_solve_check.py [grid=100] [nev=20] [tol=1e-8]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
M = ks.matrices
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nev = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-8
n = m ** 3
op = ks.csr_operator(M.to_scipy(*M.laplace3d_csr(m, m, m), n))
exact = M.laplace3d_eigs(m, m, m, nev)
for sstep in (20, 10, 0):
    ws = ks.ArnoldiWorkspace(M.start_vector(n), 40)
    ws.set_sstep(sstep)
    t0 = time.perf_counter()
    dec, hist = ks.partialschur_(op, ws, nev=nev, which="SR", tol=tol, restarts=2000)
    dt = time.perf_counter() - t0
    res, orth = ws.residual_norms(op, hist.nconverged)
    ev = np.sort(np.array(dec.eigenvalues).real)[:nev]
    err = float(np.abs(ev - exact[: len(ev)]).max()) if hist.nconverged >= nev else float("nan")
    info, rel = ws.sstep_info, ws.relation_info
    print(f"grid {m}^3 sstep {sstep:2d}: {hist} | restarts {hist.restarts} | {dt:.3f} s | ||AQ-QR||_F {res:.2e} orth {orth:.2e} "
          f"| max eigenvalue error {err:.2e} | blocks {info['blocks']} abandoned {info['abandoned']} in force {info['s']} | relation breaks {rel['breaks']}", flush=True)
    ws.close()
