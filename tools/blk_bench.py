"""Times the two streaming kernels of the s-step expansion (k_bdots / k_bupdate) stand-alone at the headline shape.
   python tools/blk_bench.py [grid=216] [maxdim=40]
BLK_DBGS: probe flags of the second-pass kernel (1: no stores, 4: cacheable instead of non-temporal stores).
Prints ms per launch and GB/s on the bytes the launch must move: 8 n (k + s) (pass 1), 8 n (k + 2 s) (pass 2)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
from arnoldimethod_jl_amd import _lib  # noqa: E402


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 216
    maxdim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    n = m ** 3
    ws = ks.ArnoldiWorkspace(n, maxdim, np.float64)
    L = _lib.load()
    shapes = [(21, 5), (26, 5), (31, 5), (36, 5), (21, 10), (31, 10), (21, 20), (21, 8), (29, 8), (25, 8), (33, 8), (21, 4), (37, 4), (21, 2), (39, 2)]
    shapes = [(k, s) for k, s in shapes if k + s <= maxdim + 1]
    if os.environ.get("BLK_SHAPES"):   # e.g. BLK_SHAPES=31:12,43:12,55:6,50:8 (k:s pairs; run-time block sizes, wide bases)
        shapes = [tuple(int(v) for v in x.split(":")) for x in os.environ["BLK_SHAPES"].split(",")]
        shapes = [(k, s) for k, s in shapes if k + s <= maxdim + 1]
    if os.environ.get("BLK_S"):
        shapes = [(k, s) for k, s in shapes if s in [int(x) for x in os.environ["BLK_S"].split(",")]]
    dbgs = [int(x) for x in os.environ.get("BLK_DBGS", "0,1,4").split(",")]
    for k, s in shapes:
        for which, name in ((0, "bdots"), (1, "bupdate")):
            dbgs0 = [int(x) for x in os.environ.get("BLK_DBGS0", "0").split(",")]
            for dbg in (dbgs if which == 1 else dbgs0):
                ms, grid = C.c_double(), C.c_int()
                _lib.check(L.ks_debug_blk_time(ws._h, k, s, which, 10, dbg, C.byref(ms), C.byref(grid)))
                b = 8.0 * n * (k + s) if which == 0 else 8.0 * n * (k + (1 if dbg & 1 else 2) * s)
                print(f"k={k:2d} s={s:2d} {name:8s} dbg={dbg} grid={grid.value:4d}  {ms.value * 1e3:8.1f} us  {b / ms.value / 1e6:7.0f} GB/s  ({b / 1e9:.2f} GB)", flush=True)


if __name__ == "__main__":
    main()
