#!/usr/bin/env python3
"""Whole solves through the library's own driver (ks_partialschur: no Python between the restarts), for traces.
    python tools/run_solver.py cfg2|headline [restarts]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
M = ks.matrices
m = 100 if (sys.argv[1] if len(sys.argv) > 1 else "cfg2") == "cfg2" else 216
restarts = int(sys.argv[2]) if len(sys.argv) > 2 else 25
n = m ** 3
op = ks.csr_operator(M.to_scipy(*M.laplace3d_csr(m, m, m), n))
ws = ks.ArnoldiWorkspace(M.start_vector(n), 40)
dec, hist = ks.partialschur_(op, ws, nev=20, which="SR", restarts=restarts)
print(hist, f"| restarts {hist.restarts} | expand {hist.seconds_expand:.4f} s host {hist.seconds_host:.4f} s rotate {hist.seconds_rotate:.4f} s "
      f"| host step {1e6 * hist.seconds_host / max(1, hist.restarts):.1f} us per restart")
