"""Per-class kernel times of the bench workload for workspaces created after growing amounts of other
allocations: shows the performance plateaus that the physical placement of V causes (run with
KS_PLACE_TRIALS=1 to see them; the default placement tuning removes them)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from __graft_entry__ import import_package
ks = import_package()
import torch
m = 216
n = m**3
ip, ix, dv = ks.matrices.laplace3d_csr(m, m, m)
A = ks.matrices.to_scipy(ip, ix, dv, n)
ctx = ks.Context(0)
op = ks.csr_operator(A, ctx)
v1 = ks.matrices.start_vector(n)
junk = []
def trial(tag):
    ws = ks.ArnoldiWorkspace(n, 40, np.float64, ctx=ctx)
    ws.reinitialize(0, v1)
    ws.iterate_arnoldi(op, 1, 20)
    k, active = 20, 0
    for it in range(4):
        if it == 1:
            ctx.profile_reset(); ctx.profile_enable(True)
        ws.iterate_arnoldi(op, k + 1, 40)
        r = ws.restart(active, 20, "SR", 1e-8, 20, 40)
        k, active = r["k"], r["nlock"]
    ctx.synchronize()
    p = ctx.profile_get(); ctx.profile_enable(False)
    print(tag, {c: round(p[c]["ms"] / p[c]["count"] * 1e3, 1) for c in ("spmv", "dots", "fused", "axpy")}, flush=True)
    ws.close()
for t in range(3):
    trial(f"same-placement {t}")
for t in range(5):
    junk.append(torch.empty(int((0.37 + 0.61 * t) * 2**30), dtype=torch.uint8, device="cuda"))
    trial(f"after junk {t} ({junk[-1].numel()/2**30:.2f} GiB kept)")
