#!/usr/bin/env python3
"""Would column blocking help config 3's SpMV?  The hashed nonsymmetric matrix (n = 1e6) split into NB column blocks, each
uploaded as its own CSR operator and timed (y overwritten, not accumulated: a lower bound of the blocked kernel's cost)."""
import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package
ks = import_package(); M = ks.matrices
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
A = M.hashed_nonsymmetric_csr(n, seed=7).tocsc()
ctx = ks.Context(0)
ws = ks.ArnoldiWorkspace(n, 2, np.float64, ctx=ctx)
ws.set_col(0, M.start_vector(n))
def timeit(op, reps=40):
    for _ in range(3): ws.apply(op, 0, 1)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(reps): ws.apply(op, 0, 1)
    p = ctx.profile_get()["spmv"]; ctx.profile_enable(False)
    return 1e3 * p["ms"] / p["count"]
os.environ["KS_SPMV_FORMAT"] = "csr"
full = ks.csr_operator(A.tocsr(), ctx)
print(f"n={n} nnz={A.nnz}: whole matrix {timeit(full):.1f} us")
for nb in (2, 4, 8):
    edges = np.linspace(0, n, nb + 1).astype(int)
    tot = 0.0; parts = []
    for b in range(nb):
        sub = sp.csc_matrix((n, n))
        B = A[:, edges[b]:edges[b + 1]]
        Bfull = sp.hstack([sp.csc_matrix((n, edges[b])), B, sp.csc_matrix((n, n - edges[b + 1]))]).tocsr()
        op = ks.csr_operator(Bfull, ctx)
        t = timeit(op); tot += t; parts.append(round(t, 1)); op.close()
    print(f"  {nb} column blocks: {parts} us, sum {tot:.1f} us")
