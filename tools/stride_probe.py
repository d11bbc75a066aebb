#!/usr/bin/env python3
"""Effect of the column stride of V on the streaming kernels: restart cycles of the headline state machine on an m^3
Laplacian (nev 20, 20/40), per-class GB/s from the library's HIP events.  KS_LD_PAD (512-byte units) is read by the library.
    KS_LD_PAD=33 python tools/stride_probe.py 216"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
M = ks.matrices
m = int(sys.argv[1]) if len(sys.argv) > 1 else 216
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = m ** 3
ctx = ks.Context(0)
op = ks.csr_operator(M.to_scipy(*M.laplace3d_csr(m, m, m), n), ctx)
ws = ks.ArnoldiWorkspace(n, 40, np.float64, ctx=ctx)
ws.reinitialize(0, M.start_vector(n))
ws.iterate_arnoldi(op, 1, 20)
k, active = 20, 0
for _ in range(2):
    r = ws.expand_restart(op, k, active, 20, "SR", 1.5e-8, 20, 40)
    k, active = r["k"], min(r["nlock"], 19)
ctx.synchronize()
t0 = time.perf_counter()
steps = 0
for _ in range(cycles):
    r = ws.expand_restart(op, k, active, 20, "SR", 1.5e-8, 20, 40)
    steps += r["steps"]
    k, active = r["k"], min(r["nlock"], 19)
ctx.synchronize()
el = time.perf_counter() - t0
ctx.profile_reset()
ctx.profile_enable(True)
for _ in range(3):
    r = ws.expand_restart(op, k, active, 20, "SR", 1.5e-8, 20, 40)
    k, active = r["k"], min(r["nlock"], 19)
p = ctx.profile_get()
ctx.profile_enable(False)
gb = {c: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9) for c, v in p.items() if v["ms"] > 0 and v["bytes"] > 0}
print(f"m={m} n={n} pad={os.environ.get('KS_LD_PAD', '0'):>5} stride mod 32K = 0x{((n + 63) // 64 * 64 + 64 * int(os.environ.get('KS_LD_PAD', '0'))) * 8 % 32768:04x}  {steps / el:8.1f} it/s  {gb}")
