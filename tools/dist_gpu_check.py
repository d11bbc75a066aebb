#!/usr/bin/env python3
"""Multi-process check of the distributed GPU path.  On a multi-GPU node: one rank per GPU.  On a
single-GPU box it can only run if RCCL accepts several ranks on one device (KS_SAME_DEVICE=1).
Each rank solves its slab of a 3-D Laplacian; rank 0 compares eigenvalues with the analytic spectrum
and checks the device-side residual."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
from arnoldimethod_jl_amd import dist as ksd  # noqa: E402


def main():
    rank = int(os.environ["RANK"])
    local_rank = 0 if os.environ.get("KS_SAME_DEVICE") == "1" else int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local_rank)
    dist.init_process_group("gloo")  # rendezvous only; the solver's collectives are RCCL inside the library
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    mx, my, mz = m, m + 1, m + 2 * dist.get_world_size()
    n = mx * my * mz
    offs = ksd.partition_rows(n, dist.get_world_size(), granule=mx * my)
    r0, r1 = int(offs[rank]), int(offs[rank + 1])
    ip, ix, dv = ks.matrices.laplace3d_csr(mx, my, mz, r0, r1, index_dtype=np.int64)
    plan = ksd.build_halo_plan(ix, offs, rank, dist)
    from arnoldimethod_jl_amd import api

    ctx = ksd.make_context(api, dist, local_rank)
    op = ksd.dist_operator(api, ctx, ip, dv, plan, n)
    ws = api.ArnoldiWorkspace(r1 - r0, 30, np.float64, ctx=ctx, n_global=n, row_begin=r0)
    ws._v1 = ks.matrices.start_vector(r1 - r0, row_begin=r0)
    F, hist = ks.partialschur_(op, ws, nev=6, which="SR", tol=1e-10, mindim=12, maxdim=30, restarts=300)
    res, orth = ws.residual_norms(op, F.nconverged)
    exact = ks.matrices.laplace3d_eigs(mx, my, mz, 6)
    err = np.abs(np.sort(F.eigenvalues.real)[:6] - exact).max() if F.nconverged >= 6 else float("nan")
    ok = hist.converged and res < 1e-8 and orth < 1e-12 and err < 1e-8
    print(f"[rank {rank}] {hist} resid={res:.2e} orth={orth:.2e} eig_err={err:.2e} rows {r0}:{r1} -> {'OK' if ok else 'FAIL'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
