"""The out-of-place lottery of k_bupdate_mfma (profiles/r06_bupdate_lottery.txt): the second pass of the headline's block (21 + 20
columns read, 20 written) timed stand-alone in ONE process,
   in place            the block is read from the basis columns it is written to
   scratch             the block is read from the scratch columns of the fused rotation (what a headline cycle does)
   scratch, re-rolled  ... after giving the scratch columns a FRESH allocation (the new one is made before the old one is freed), 8 times
and the same for a second workspace of the process (a fresh allocation of the BASIS).  Is it the scratch columns' placement, the
basis', or neither?    python tools/bupdate_lottery.py [grid=216]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
from arnoldimethod_jl_amd import _lib  # noqa: E402


def t(L, ws, dbg, reps=10):
    ms, grid = C.c_double(), C.c_int()
    _lib.check(L.ks_debug_blk_time(ws._h, 21, 20, 1, reps, dbg, C.byref(ms), C.byref(grid)))
    return ms.value * 1e3


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 216
    n = m ** 3
    L = _lib.load()
    keep = []
    for w in range(3):
        ws = ks.ArnoldiWorkspace(n, 40, np.float64)
        keep.append(ws)   # (held: the next workspace gets other pages)
        line = [f"workspace {w}: in place {t(L, ws, 0):6.1f} us | scratch {t(L, ws, 256):6.1f} | re-rolled"]
        for _ in range(8):
            line.append(f"{t(L, ws, 256 | 512):6.1f}")
        line.append(f"| again without re-rolling {t(L, ws, 256):6.1f} {t(L, ws, 256):6.1f}")
        print(" ".join(line), flush=True)


if __name__ == "__main__":
    main()
