#!/usr/bin/env python3
"""Multi-process check of the distributed GPU path (also driven by tests/test_gpu_parity.py).

On a multi-GPU node: one rank per GPU.  On a single-GPU box: KS_SAME_DEVICE=1 KS_TRANSPORT=p2p puts all
ranks on device 0 (RCCL refuses several ranks per device, the peer-to-peer transport does not care).

    KS_SAME_DEVICE=1 KS_TRANSPORT=p2p python -m torch.distributed.run --nproc-per-node 2 \\
        --master-addr 127.0.0.1 --master-port 29517 tools/dist_gpu_check.py MODE [m]

MODE
  laplace  every rank solves its slab of an m x (m+1) x (m+2w) Laplacian (ghosts = whole grid planes, sent
           as contiguous runs); eigenvalues vs the analytic spectrum, device-side residual, and rank 0
           repeats the solve on a single-GPU context: same mvproducts, same Ritz values.
  hashed   nonsymmetric matrix with hashed columns (every rank is everybody's neighbour, scattered
           send lists); distributed vs single-GPU run: same mvproducts, same Ritz values, ||AQ - QR||.
  wide     the same matrix with maxdim = 60 > 40: the eager (un-fused) DGKS sequence with stand-alone
           reductions (k_p2p_allreduce), chunked inner products, out-of-place rotation.
  complex  ComplexF64 variant (complex diagonal shift): 16-byte elements through halo and reductions.
  timeout  rank 1 never joins; rank 0 must get CommTimeout (KS_ERR_COMM) after KS_P2P_TIMEOUT_S
           seconds instead of hanging.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
from arnoldimethod_jl_amd import _lib, api, dist as ksd  # noqa: E402


def main():
    rank = int(os.environ["RANK"])
    local_rank = 0 if os.environ.get("KS_SAME_DEVICE") == "1" else int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local_rank)
    dist.init_process_group("gloo")  # rendezvous only; the solver's exchanges happen inside the library
    world = dist.get_world_size()
    mode = sys.argv[1] if len(sys.argv) > 1 else "laplace"
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctx = ksd.make_context(api, dist, local_rank)

    if mode == "timeout":
        ok = True
        if rank == 0:
            try:
                wsx = api.ArnoldiWorkspace(64, 4, np.float64, ctx=ctx, n_global=64 * world, row_begin=0)
                wsx.set_col(0, np.ones(64))
                wsx.norm(0)  # an all-reduce nobody else joins
                ok = False
            except _lib.CommTimeout as e:
                print(f"[rank 0] got the expected CommTimeout: {e}", flush=True)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)

    if mode == "laplace":
        mx, my, mz = m, m + 1, m + 2 * world
        n = mx * my * mz
        offs = ksd.partition_rows(n, world, granule=mx * my)
        r0, r1 = int(offs[rank]), int(offs[rank + 1])
        ip, ix, dv = ks.matrices.laplace3d_csr(mx, my, mz, r0, r1, index_dtype=np.int64)
        kw = dict(nev=6, which="SR", tol=1e-10, mindim=12, maxdim=30, restarts=300)
        full = lambda: sp.csr_matrix(ks.matrices.laplace3d_csr(mx, my, mz, 0, n, index_dtype=np.int64)[::-1], shape=(n, n))  # noqa: E731
    else:
        n = m * m * m
        diag = np.linspace(1.0, 40.0, n) + (0.5j * np.cos(np.arange(n)) if mode == "complex" else 0.0)
        A = (ks.matrices.hashed_nonsymmetric_csr(n, seed=11) + sp.diags(diag)).tocsr()
        A.sort_indices()
        offs = ksd.partition_rows(n, world)
        r0, r1 = int(offs[rank]), int(offs[rank + 1])
        B = A[r0:r1]
        ip, ix, dv = B.indptr.astype(np.int64), B.indices.astype(np.int64), B.data
        kw = dict(nev=5, which="LR", tol=1e-10, mindim=10, maxdim=25, restarts=300)
        if mode == "wide":
            kw = dict(nev=12, which="LR", tol=1e-10, mindim=30, maxdim=60, restarts=300)
        full = lambda: A  # noqa: E731
    dtype = np.complex128 if mode == "complex" else np.float64

    plan = ksd.build_halo_plan(ix, offs, rank, dist)
    op = ksd.dist_operator(api, ctx, ip, dv, plan, n)
    ws = api.ArnoldiWorkspace(r1 - r0, kw["maxdim"], dtype, ctx=ctx, n_global=n, row_begin=r0)
    ws._v1 = ks.matrices.start_vector(r1 - r0, row_begin=r0).astype(dtype)
    F, hist = ks.partialschur_(op, ws, **kw)
    res, orth = ws.residual_norms(op, F.nconverged)
    # ||A Q - Q R||_F over nconverged columns; the solver's criterion is per vector, tol * |lambda| (src/run.jl:330)
    ok = bool(hist.converged and res < 10 * kw["tol"] * np.abs(F.eigenvalues).max() * max(1, F.nconverged) and orth < 1e-12)
    msg = f"{hist} resid={res:.2e} orth={orth:.2e} neighbours={len(plan.neigh)} rows {r0}:{r1}"
    if mode == "laplace":
        exact = ks.matrices.laplace3d_eigs(mx, my, mz, 6)
        err = np.abs(np.sort(F.eigenvalues.real)[:6] - exact).max() if F.nconverged >= 6 else float("nan")
        ok = ok and err < 1e-8
        msg += f" eig_err={err:.2e}"
    if rank == 0:  # same problem on one GPU: the row partition must not change what the solver does
        ws1 = ks.ArnoldiWorkspace(n, kw["maxdim"], dtype)
        ws1._v1 = ks.matrices.start_vector(n).astype(dtype)
        F1, h1 = ks.partialschur_(full(), ws1, **kw)
        dv_ = np.abs(np.sort_complex(F1.eigenvalues) - np.sort_complex(F.eigenvalues)).max()
        same = h1.mvproducts == hist.mvproducts and dv_ < 1e-9
        print(f"[rank 0] single-GPU run: {h1}; max Ritz value difference {dv_:.2e}; same: {same}", flush=True)
        ok = ok and same
    print(f"[rank {rank}] {msg} -> {'OK' if ok else 'FAIL'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
