#!/usr/bin/env python3
"""Single-GPU probe of the distributed code path's fixed costs: the same slab problem solved (a) on a plain
context and (b) on a 1-rank RCCL context whose halo plan exchanges two planes with itself (so the pack kernel,
the ncclSend/ncclRecv group, the reduce-only / all-reduce / post kernel variants all run) and (c) on a 1-rank
peer-to-peer context (csrc/ks_p2p.hpp).  Prints ms per Arnoldi iteration; the differences are what the
multi-GPU structure costs before any real link latency."""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
from arnoldimethod_jl_amd import api, dist as ksd  # noqa: E402


def run(ctx, op, n, label, cycles=10):
    ws = ks.ArnoldiWorkspace(n, 40, np.float64, ctx=ctx)
    ws.reinitialize(0, ks.matrices.start_vector(n))
    ws.iterate_arnoldi(op, 1, 20)
    k, active = 20, 0
    for it in range(cycles + 2):
        if it == 2:
            ctx.synchronize()
            t0 = time.perf_counter()
            steps = 0
        ws.iterate_arnoldi(op, k + 1, 40)
        if it >= 2:
            steps += 40 - k
        r = ws.restart(active, 20, "SR", 1e-8, 20, 40)
        k, active = r["k"], r["nlock"]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:28s} {1e3 * dt / steps:.4f} ms/iter  ({steps / dt:.0f} iters/s)", flush=True)


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 108
    legs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["plain", "rccl", "p2p"]
    mx = my = m
    mz = m
    n = mx * my * mz
    ip, ix, dv = ks.matrices.laplace3d_csr(mx, my, mz, index_dtype=np.int64)
    # a plan that sends the first and last plane to this very rank (values unused: the ghost columns are
    # referenced by nobody, so the operator is unchanged but every exchange step executes)
    plane = mx * my
    send_idx = np.concatenate([np.arange(plane), np.arange(n - plane, n)]).astype(np.int32)
    plan = ksd.HaloPlan(n_local=n, nghost=2 * plane, neigh=np.array([0], dtype=np.int32), send_ptr=np.array([0, 2 * plane], dtype=np.int64),
                        send_idx=send_idx, recv_cnt=np.array([2 * plane], dtype=np.int64), ghost_global=np.zeros(0), colidx_local=ix.astype(np.int32))
    # (memory placement differs between allocations of one process by a few per cent: compare legs run in
    # SEPARATE processes -- `dist_overhead.py 108 plain`, `... rccl`, `... p2p`)
    for leg in legs:
        if leg == "plain":
            ctx = ks.Context(0)
            op = ks.csr_operator(ks.matrices.to_scipy(ip, ix.astype(np.int32), dv, n), ctx)
        elif leg == "rccl":   # pack kernel / ncclSend+ncclRecv group, reduce-only -> all-reduce -> post kernels
            ctx = ks.Context(0, 0, 1, ks.Context.unique_id())
            op = ksd.dist_operator(api, ctx, ip, dv, plan, n)
        else:                 # one push kernel, exchange folded into the reduction kernels
            ctx = ks.Context(0, 0, 1, p2p=True)
            op = ksd.dist_operator(api, ctx, ip, dv, plan, n)
        run(ctx, op, n, f"{leg:5s} context n={n}")
        del op, ctx


if __name__ == "__main__":
    main()
