#!/usr/bin/env python3
"""The random dominant-outlier sweep of tests/test_gpu_random_stress.py (_outlier_case, 14 seeds) with the in-chain deflation on and
off: blocks completed / abandoned, block size in force at the end, products (oracle's in brackets), wall time of the solve.
    python tools/outlier_sweep.py        (profiles/r06c_outlier_sweep.txt)"""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import import_package  # noqa: E402
from oracle import arnoldi as oa  # noqa: E402

ks = import_package()
src = open(os.path.join(ROOT, "tests", "test_gpu_random_stress.py")).read().replace("pkg = import_package()", "pkg = None")
m = types.ModuleType("t")
exec(compile(src, "t", "exec"), m.__dict__)

tot = {"1": [0, 0, 0.0], "0": [0, 0, 0.0]}
for seed in range(14):
    A, v1, kw, exact, s_blk = m._outlier_case(seed)
    ref, rh = oa.partialschur(A, v1=v1, **kw)
    row = f"seed {seed:2d} n={A.shape[0]:5d} {str(A.dtype):10s} s={s_blk:2d} {kw['which']} nev {kw['nev']:2d} {kw['mindim']}/{kw['maxdim']} outliers {len(exact)} |"
    for on in ("1", "0"):
        os.environ["KS_CHAIN_DEFLATE"] = on
        op = ks.csr_operator(A)
        best = None
        for rep in range(2):
            ws = ks.ArnoldiWorkspace(A.shape[0], kw["maxdim"], A.dtype)
            ws.set_sstep(s_blk)
            ws._v1 = v1
            ws.ctx.synchronize()
            t0 = time.perf_counter()
            dec, h = ks.partialschur_(op, ws, **kw)
            ws.ctx.synchronize()
            dt = time.perf_counter() - t0
            info = ws.sstep_info
            ws.close()
            best = (dt, h, info)
        dt, h, info = best
        tot[on][0] += info["abandoned"]; tot[on][1] += int(info["s"] < s_blk); tot[on][2] += dt
        row += f" deflate={on}: blocks {info['blocks']:4d} abandoned {info['abandoned']:2d} s at the end {info['s']:2d} products {h.mvproducts} ({rh.mvproducts}) {1e3 * dt:7.1f} ms |"
    print(row, flush=True)
for on in ("1", "0"):
    print(f"deflate={on}: {tot[on][0]} blocks abandoned over the sweep, {tot[on][1]} of 14 runs end with a lowered block size, {1e3 * tot[on][2]:.0f} ms in total")
