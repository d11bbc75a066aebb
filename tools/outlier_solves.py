#!/usr/bin/env python3
"""Whole solves of :LM problems with dominant outliers (round 6: in-chain deflation, csrc/ks_block_kernels.hpp kDeflMax):
BASELINE config 3's matrix at full size with three planted eigenvalues 10x the bulk, the same without planting, and the operator of
test/partial_schur.jl:122-138 (disc + outlier) as a dense 4 000 x 4 000 matrix -- library default, KS_CHAIN_DEFLATE=0 (rounds 3-5)
and step by step.  Prints wall time of the solve, products, block statistics, ||AQ - QR|| on the device.
    python tools/outlier_solves.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()


def solve(label, mk_op, v1, wsdim, variants, **kw):
    for name, env, sstep in variants:
        for k_, v_ in env.items():
            os.environ[k_] = v_
        op = mk_op()
        best = None
        for rep in range(2):   # (second run: allocations and code objects warm)
            ws = ks.ArnoldiWorkspace(v1, wsdim)
            if sstep is not None:
                ws.set_sstep(sstep)
            ws.ctx.synchronize()
            t0 = time.perf_counter()
            F, hist = ks.partialschur_(op, ws, **kw)
            ws.ctx.synchronize()
            dt = time.perf_counter() - t0
            info = ws.sstep_info
            dres, dorth = F.workspace.residual_norms(op, F.nconverged)
            best = (dt, hist, info, dres, dorth)
            ws.close()
        dt, hist, info, dres, dorth = best
        print(f"{label:28s} {name:22s} {1e3 * dt:9.1f} ms  products {hist.mvproducts:5d}  converged {hist.nconverged:2d}  blocks {info['blocks']:4d} abandoned {info['abandoned']} "
              f"s in force {info['s']:2d} deflated blocks {info['deflated_blocks']:4d} (columns {info['deflated_columns']})  ||AQ-QR|| {dres:.1e}  ||Q'Q-I|| {dorth:.1e}", flush=True)
        for k_ in env:
            os.environ.pop(k_, None)


VARIANTS = [("default", {}, None), ("KS_CHAIN_DEFLATE=0", {"KS_CHAIN_DEFLATE": "0"}, None), ("step by step", {}, 0)]

if __name__ == "__main__":
    n = 1_000_000
    # (three planted eigenvalues 10x the bulk, one of them a conjugate pair: locked at the first restarts, the remaining six wanted
    # ones are the bulk's largest)
    planted = [(30.0, 0.0), (25.0, 10.0), (-28.0, 0.0)]
    A = ks.matrices.hashed_nonsymmetric_csr(n, seed=7, planted=planted)
    solve("config 3 + dominant outliers", lambda: ks.csr_operator(A), ks.matrices.start_vector(n), 30, VARIANTS, nev=6, which="LM", tol=1e-8, mindim=10, maxdim=30, restarts=100)
    A3 = ks.matrices.hashed_nonsymmetric_csr(n, seed=7)
    solve("config 3 (no planting)", lambda: ks.csr_operator(A3), ks.matrices.start_vector(n), 20, VARIANTS, nev=10, which="LM", tol=1e-8, restarts=200)
    m = 4000
    rng = np.random.default_rng(5)
    D = rng.standard_normal((m, m)) / np.sqrt(m)
    D[0, 0] = 50.0
    solve("dense disc + outlier 4000", lambda: ks.dense_operator(D), ks.matrices.start_vector(m), 30, VARIANTS, nev=5, which="LM", tol=1e-10, mindim=10, maxdim=30, restarts=200)
