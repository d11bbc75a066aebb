#!/usr/bin/env python3
"""Why the stencil SpMV takes 57 us inside the solver and 44 us in a stand-alone loop (216^3 Laplacian): where does x come from?
Times y = A x (HIP events on the library's stream, ctx.profile) in four situations:
  warm        the same x over and over (stand-alone benchmark: x sits in the 256 MiB memory-side cache)
  flushed     512 MiB streamed through the device before every product (x and the masks must come from HBM)
  produced    x freshly written by a streaming kernel with non-temporal stores (k_copy): what the solver's chain sees
  chain       products chained column to column, as the Newton basis of the s-step expansion does
Bytes: the product streams 17 B per row (171 MB); x alone is 81 MB -- 13 us at 6.3 TB/s."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
M = ks.matrices


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 216
    n = m ** 3
    ctx = ks.Context(0)
    op = ks.csr_operator(M.to_scipy(*M.laplace3d_csr(m, m, m), n), ctx)
    ws = ks.ArnoldiWorkspace(n, 12, np.float64, ctx=ctx)
    ws.set_col(0, M.start_vector(n))
    ws.set_col(2, M.start_vector(n))
    junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")

    def timed(label, body, reps=20):
        ctx.synchronize()
        ctx.profile_reset()
        ctx.profile_enable(True)
        for i in range(reps):
            body(i)
        ctx.synchronize()
        p = ctx.profile_get()["spmv"]
        ctx.profile_enable(False)
        print(f"{label:10s} {1e3 * p['ms'] / p['count']:7.1f} us per product ({p['count']} products, layout {op.format['layout']})", flush=True)

    def warm(i):
        ws.apply(op, 0, 1)

    def flushed(i):
        ctx.synchronize()
        junk.add_(1)
        torch.cuda.synchronize()
        ws.apply(op, 0, 1)

    def produced(i):
        ws.copy_col(0, 2)      # k_copy: non-temporal stores
        ws.apply(op, 0, 1)

    def chain(i):
        j = i % 10
        ws.apply(op, j, j + 1)

    timed("(first)", warm, reps=3)   # (lazy code-object load of the first launches: not a measurement)
    timed("warm", warm)
    timed("flushed", flushed)
    timed("produced", produced)
    timed("chain", chain)


if __name__ == "__main__":
    main()
