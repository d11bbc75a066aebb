#!/usr/bin/env python3
"""How the CPU baseline (oracle/cpu_backend.cpp) scales with the OpenMP team on the box it runs on.
    OMP_NUM_THREADS=64 OMP_PROC_BIND=spread OMP_PLACES=cores python tools/cpu_baseline_probe.py [grid]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402
from oracle import cpuref  # noqa: E402

ks = import_package()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 216
n = m ** 3
A = ks.matrices.to_scipy(*ks.matrices.laplace3d_csr(m, m, m), n)
tb = cpuref.timed_cycles_csr(A, nev=20, which="SR", mindim=20, maxdim=40, cycles=2)
st = cpuref.stream_triad_gbs()
print(f"threads {tb['threads']:4d} bind={os.environ.get('OMP_PROC_BIND')} places={os.environ.get('OMP_PLACES')}: {tb['steps'] / tb['seconds']:6.2f} it/s | per step: "
      f"spmv {1e3 * tb['t_spmv'] / tb['steps']:6.1f} ms orth {1e3 * tb['t_orth'] / tb['steps']:6.1f} ms | rotation {1e3 * tb['t_rot'] / 2:6.1f} ms per restart | stream triad {st:6.0f} GB/s", flush=True)
