#!/usr/bin/env python3
"""Single-GPU probe of the distributed code path's fixed costs: the same slab problem (a z-periodic Laplacian, so that
the slab has ghost planes on both sides) solved (a) on a plain context, (b) on a 1-rank RCCL context whose halo plan
exchanges the two boundary planes with itself (pack kernel / ncclSend+ncclRecv group, reduce-only -> all-reduce -> post
kernels all run, and the boundary rows read the ghosts) and (c) on a 1-rank peer-to-peer context (csrc/ks_p2p.hpp:
exchange folded into the SpMV and into the reduction kernel).  Prints ms per Arnoldi iteration; the differences are
what the multi-GPU structure costs before any real link latency."""
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
        # one restart cycle per library call, as bench.py and ks_partialschur run it (the restart's rotation may then stay pending
        # for the next expansion's fused first pass and the first products of the next chain run behind this expansion)
        if it >= 2:
            steps += 40 - k
        r = ws.expand_restart(op, k, active, 20, "SR", 1e-8, 20, 40)
        k, active = r["k"], min(r["nlock"], 19)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    info = ws.sstep_info
    print(f"{label:28s} {1e3 * dt / steps:.4f} ms/iter  ({steps / dt:.0f} iters/s)  [blocks {info['blocks']}, fused rotations {info['fused_rotations']}, chains adopted {info['chains_adopted']}]", flush=True)


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 108
    legs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["plain", "rccl", "p2p"]
    mx = my = m
    mz = int(os.environ.get("KS_PROBE_MZ", m))
    n = mx * my * mz
    plane = mx * my
    ip, ix, dv = ks.matrices.laplace3d_csr(mx, my, mz, index_dtype=np.int64)
    import scipy.sparse as sp

    A0 = ks.matrices.to_scipy(ip, ix.astype(np.int32), dv, n)
    # PERIODIC in z, so that the slab really has two neighbours' worth of ghost planes -- here both are this very rank:
    # rows of the first plane couple to the last plane and vice versa.  Single-rank form: the wrap-around entries are
    # ordinary columns.  Distributed form: they are GHOST columns n .. n + 2 plane (slots [0, plane) = copy of the last
    # plane, [plane, 2 plane) = copy of the first), and the halo plan sends both planes to rank 0 itself -- every step of
    # the exchange (push / pack + send/recv, flags, waits of the boundary tiles, ghost reads of the SpMV) really executes
    # and the boundary rows really depend on it.
    first, last = np.arange(plane), np.arange(n - plane, n)
    wrap_plain = sp.coo_matrix((-np.ones(2 * plane), (np.concatenate([first, last]), np.concatenate([last, first]))), shape=(n, n))
    A_plain = (A0 + wrap_plain).tocsr()
    A_plain.sort_indices()
    wrap_ghost = sp.coo_matrix((-np.ones(2 * plane), (np.concatenate([first, last]), np.concatenate([n + np.arange(plane), n + plane + np.arange(plane)]))),
                               shape=(n, n + 2 * plane))
    A_dist = (sp.csr_matrix((A0.data, A0.indices, A0.indptr), shape=(n, n + 2 * plane)) + wrap_ghost).tocsr()
    A_dist.sort_indices()
    send_idx = np.concatenate([last, first]).astype(np.int32)
    plan = ksd.HaloPlan(n_local=n, nghost=2 * plane, neigh=np.array([0], dtype=np.int32), send_ptr=np.array([0, 2 * plane], dtype=np.int64),
                        send_idx=send_idx, recv_cnt=np.array([2 * plane], dtype=np.int64), ghost_global=np.zeros(0), colidx_local=A_dist.indices.astype(np.int32))
    # (memory placement differs between allocations of one process by a few per cent: compare legs run in
    # SEPARATE processes -- `dist_overhead.py 108 plain`, `... rccl`, `... p2p`)
    for leg in legs:
        if leg == "plain":
            ctx = ks.Context(0)
            op = ks.csr_operator(A_plain, ctx)
        elif leg == "rccl":   # pack kernel / ncclSend+ncclRecv group, reduce-only -> all-reduce -> post kernels
            ctx = ks.Context(0, 0, 1, ks.Context.unique_id())
            op = ksd.dist_operator(api, ctx, A_dist.indptr.astype(np.int64), A_dist.data, plan, n)
        else:                 # exchange folded into the SpMV and into the reduction kernels
            ctx = ks.Context(0, 0, 1, p2p=True)
            op = ksd.dist_operator(api, ctx, A_dist.indptr.astype(np.int64), A_dist.data, plan, n)
        if os.environ.get("KS_PROBE_CHECK", "1") == "1":  # the three forms are the same operator
            ws = ks.ArnoldiWorkspace(n, 2, np.float64, ctx=ctx)
            x = ks.matrices.start_vector(n)
            ws.set_col(0, x)
            ws.apply(op, 0, 1)
            ws.apply(op, 0, 1)
            err = np.abs(ws.col(1) - A_plain @ x).max()
            assert err < 1e-12, f"{leg}: SpMV differs from the periodic Laplacian by {err:.2e}"
            ws.close()
        run(ctx, op, n, f"{leg:5s} context n={n} layout={op.format['layout']}")
        del op, ctx


if __name__ == "__main__":
    main()
