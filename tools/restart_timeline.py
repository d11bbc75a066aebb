#!/usr/bin/env python3
"""Host-side timeline of restart cycles (ks_expand_restart's `seconds`): wall time of the expansion call (enqueue + wait
for the device), of the host's Schur / reorder / restore step, of enqueueing the rotation (T Q on the host, upload, launch),
and of the whole cycle -- at the 8-way share of the headline (108: 216 x 216 x 27 rows) or any grid.
    python tools/restart_timeline.py [m] [mz]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package
ks = import_package()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 108
mz = int(sys.argv[2]) if len(sys.argv) > 2 else m
n = m * m * mz
ip, ix, dv = ks.matrices.laplace3d_csr(m, m, mz)
ctx = ks.Context(0)
op = ks.csr_operator(ks.matrices.to_scipy(ip, ix, dv, n), ctx)
ws = ks.ArnoldiWorkspace(n, 40, np.float64, ctx=ctx)
ws.reinitialize(0, ks.matrices.start_vector(n))
ws.iterate_arnoldi(op, 1, 20)
k, active = 20, 0
rows = []
for it in range(14):
    ctx.synchronize()
    t0 = time.perf_counter()
    r = ws.expand_restart(op, k, active, 20, "SR", None, 20, 40)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    if it >= 2:
        rows.append((40 - k, *[1e6 * x for x in r["seconds"]], 1e6 * (t1 - t0), 1e6 * (t2 - t1)))
    k, active = r["k"], r["nlock"]
a = np.array(rows)
print(f"n={n}: per restart cycle (us, mean of {len(rows)}): steps {a[:,0].mean():.0f} | expansion call {a[:,1].mean():.0f} | host Schur step {a[:,2].mean():.0f} "
      f"| rotation enqueue (T Q, upload, launch) {a[:,3].mean():.0f} | call total {a[:,4].mean():.0f} | rotation still running after the call {a[:,5].mean():.0f}")
