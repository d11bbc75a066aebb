#!/usr/bin/env python3
"""Column-blocked CSR (config 3's matrix): one launch (k_spmv_csr_cb) against one launch per block, per size / block count
/ rows per workgroup.  Usage: cb_single_ab.py [n ...]"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package
ks = import_package(); M = ks.matrices

def timeit(ctx, ws, op, reps=40):
    for _ in range(3): ws.apply(op, 0, 1)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(reps): ws.apply(op, 0, 1)
    p = ctx.profile_get()["spmv"]; ctx.profile_enable(False)
    return 1e3 * p["ms"] / p["count"]

for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 2_000_000, 10_000_000]:
    A = M.hashed_nonsymmetric_csr(n, seed=7)
    ctx = ks.Context(0)
    ws = ks.ArnoldiWorkspace(n, 2, np.float64, ctx=ctx)
    x = M.start_vector(n)
    ws.set_col(0, x)
    ref = A @ x
    rows = []
    for env in ({"KS_SPMV_COLBLOCKS": "0"}, {"KS_SPMV_CB_SINGLE": "0"}, {}, {"KS_SPMV_CB_RPT": "1"}, {"KS_SPMV_CB_RPT": "2"}, {"KS_SPMV_CB_RPT": "4"},
                {"KS_SPMV_CB_RPT": "8"}, {"KS_SPMV_CB_RPT": "16"}):
        for k in ("KS_SPMV_COLBLOCKS", "KS_SPMV_CB_SINGLE", "KS_SPMV_CB_RPT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        op = ks.csr_operator(A, ctx)
        t = timeit(ctx, ws, op)
        same = bool((ws.col(1) == ref).all()) or float(np.abs(ws.col(1) - ref).max())
        rows.append((env or "default", op.format["layout"], round(t, 1), same))
        op.close()
    print(f"n={n} nnz={A.nnz}")
    for r in rows:
        print("   ", r, flush=True)
    ws.close(); ctx.close()
