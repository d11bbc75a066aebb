#!/usr/bin/env python3
"""Multi-process check of the distributed GPU path (also driven by tests/test_gpu_parity.py).

On a multi-GPU node: one rank per GPU.  On a single-GPU box: KS_SAME_DEVICE=1 KS_TRANSPORT=p2p|host puts all
ranks on device 0 (RCCL refuses several ranks per device; the peer-to-peer transport does not care, and the
host-staged transport runs the RCCL transport's launch structure with the exchanges going through gloo).

    KS_SAME_DEVICE=1 KS_TRANSPORT=p2p python -m torch.distributed.run --nproc-per-node 2 \\
        --master-addr 127.0.0.1 --master-port 29517 tools/dist_gpu_check.py MODE [m]

MODE
  laplace  every rank solves its slab of an m x (m+1) x (m+2w) Laplacian (ghosts = whole grid planes, sent
           as contiguous runs); eigenvalues vs the analytic spectrum, device-side residual, and rank 0
           repeats the solve on a single-GPU context: same mvproducts, same Ritz values.
  hashed   nonsymmetric matrix with hashed columns (every rank is everybody's neighbour, scattered
           send lists); distributed vs single-GPU run: same mvproducts, same Ritz values, ||AQ - QR||.
  wide     the same matrix with maxdim = 60: fused path at 15 columns per wave, inner products in two launches.
  eager    maxdim = 70 > 64: the eager (un-fused) DGKS sequence with stand-alone reductions, chunked inner
           products, out-of-place rotation.
  complex  ComplexF64 variant (complex diagonal shift): 16-byte elements through halo and reductions.
  outlier  the hashed matrix with three planted eigenvalues 10x the bulk (one of them a conjugate pair), :LM: the in-chain
           deflation of the block expansion with its dot products all-reduced over the ranks (KS_CHECK_BLOCKS=1 also asks for
           deflated blocks and for no abandoned one).
  shard5   BASELINE config 5: m^3 Laplacian (m = 464 with 8 ranks = the true per-rank size 464 x 464 x 58), nev = 20,
           mindim/maxdim = 20/40, two restart cycles: same restart trail and Ritz values as rank 0's single-process
           run of the whole problem, Arnoldi relation and orthogonality evaluated on the device for both.
  timeout  rank 1 never joins; rank 0 must get CommTimeout (KS_ERR_COMM) after KS_P2P_TIMEOUT_S
           seconds instead of hanging.
  cbprod   hashed matrix at n = m^3 (m = 100: large enough for the automatic column blocks): every rank's product in the
           column-blocked layout (local blocks + ghost blocks) against the single-GPU product, bit for bit
  halo / halohashed
           stress of the ghost exchange alone (round 3: folded into the SpMV launch in peer-to-peer mode): 200 rounds of
           CHAINS of 1-4 back-to-back products y = A^c x on the slab Laplacian / the hashed matrix, fresh x every round, no
           host synchronisation inside a chain (consecutive exchanges alternate the ghost slots), every result compared
           with the same chain on the whole matrix on the host.  A stale or early-read ghost entry is an O(1) error.
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


def setup_skew(rank):
    """KS_TEST_SETUP_SKEW_S="<rank>:<seconds>": that rank dawdles during set-up (stands for a slow host-side assembly: 8
    processes on a 16-CPU quota).  With the control-plane barrier between set-up and the first exchange
    (dist.ready_barrier) the others simply wait for it; without it their exchange kernels would give up after
    KS_P2P_TIMEOUT_S."""
    spec = os.environ.get("KS_TEST_SETUP_SKEW_S", "")
    if spec:
        import time

        r, sec = spec.split(":")
        if int(r) == rank:
            time.sleep(float(sec))


def run_cycles(op, ws, v1, cycles, nev=20, mindim=20, maxdim=40):
    """bench.py's state machine (src/run.jl:267-368): initial expansion, then `cycles` restart cycles.  Returns the
    restart trail [(k, nlock)], the Ritz values entering the last restart and the device-side invariants."""
    tol = float(np.sqrt(np.finfo(np.float64).eps))
    ws.reinitialize(0, v1)
    ws.iterate_arnoldi(op, 1, mindim)
    k, active, trail, ritz, steps = mindim, 0, [], None, mindim
    for _ in range(cycles):
        ws.iterate_arnoldi(op, k + 1, maxdim)
        steps += maxdim - k
        r = ws.restart(active, nev, "SR", tol, mindim, maxdim)
        k, active = r["k"], r["nlock"]
        trail.append((k, active))
        ritz = np.sort_complex(r["eigenvalues"][:k])
    H = np.array(ws.H)
    rel, orth = ws.arnoldi_relation(op, k)  # ||A V_k - V_{k+1} H_k||_F and ||V'V - I||_F on the device (collective)
    return dict(trail=trail, ritz=ritz, steps=steps, rel=rel, orth=orth, hnorm=float(np.linalg.norm(H[: k + 1, :k])), k=k)


def shard5(rank, world, ctx, m, cycles=2):
    """BASELINE config 5 at TRUE per-rank size when m = 464 and world = 8 (464 x 464 x 58 rows per rank, 4.1 GB of
    basis each): the row-partitioned run must walk the same restart trail and find the same Ritz values as the
    single-process run of the whole m^3 problem (rank 0 repeats it: V = 33 GB on one GPU), and both must satisfy the
    reference's two invariants (test/expansion.jl:29-30) evaluated on the device."""
    import time

    n = m ** 3
    offs = ksd.partition_rows(n, world, granule=m * m)
    r0, r1 = int(offs[rank]), int(offs[rank + 1])
    t0 = time.time()
    ip, ix, dv = ks.matrices.laplace3d_csr(m, m, m, r0, r1, index_dtype=np.int64)
    plan = ksd.build_halo_plan(ix, offs, rank, dist)
    op = ksd.dist_operator(api, ctx, ip, dv, plan, n)
    del ip, ix, dv
    ws = api.ArnoldiWorkspace(r1 - r0, 40, np.float64, ctx=ctx, n_global=n, row_begin=r0)
    setup_skew(rank)
    ksd.ready_barrier(dist)  # the spin budget of the first exchange must not pay for the slowest rank's assembly
    d = run_cycles(op, ws, ks.matrices.start_vector(r1 - r0, row_begin=r0), cycles)
    t1 = time.time()
    ok = d["rel"] <= 1e-12 * d["hnorm"] * 10 and d["orth"] <= np.sqrt(np.finfo(np.float64).eps) / 100
    msg = (f"rows {r0}:{r1} ({r1 - r0}) trail={d['trail']} steps={d['steps']} rel={d['rel']:.2e} (|H|={d['hnorm']:.2e}) "
           f"orth={d['orth']:.2e} [{t1 - t0:.1f} s]")
    ws.close()
    op.close()
    if rank == 0:
        t0 = time.time()
        A = ks.matrices.to_scipy(*ks.matrices.laplace3d_csr_chunked(m, m, m), n)
        op1 = ks.csr_operator(A)
        del A
        t1 = time.time()
        ws1 = ks.ArnoldiWorkspace(n, 40, np.float64)
        s = run_cycles(op1, ws1, ks.matrices.start_vector(n), cycles)
        t2 = time.time()
        dv_ = float(np.abs(s["ritz"] - d["ritz"]).max()) if s["ritz"].shape == d["ritz"].shape else float("nan")
        ok1 = s["rel"] <= 1e-12 * s["hnorm"] * 10 and s["orth"] <= np.sqrt(np.finfo(np.float64).eps) / 100
        same = s["trail"] == d["trail"] and s["steps"] == d["steps"] and dv_ <= 1e-9 * max(1.0, float(np.abs(s["ritz"]).max()))
        print(f"[rank 0] single-GPU run of the whole {m}^3 problem (n={n}, V={8 * n * 41 / 1e9:.1f} GB): trail={s['trail']} "
              f"rel={s['rel']:.2e} (|H|={s['hnorm']:.2e}) orth={s['orth']:.2e} invariants: {ok1}; assembly+upload {t1 - t0:.1f} s, "
              f"{s['steps']} iterations {t2 - t1:.1f} s; max Ritz value difference {dv_:.2e}; same: {same}", flush=True)
        ok = ok and ok1 and same
        ws1.close()
        op1.close()
    print(f"[rank {rank}] {msg} -> {'OK' if ok else 'FAIL'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


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

    if mode == "shard5":
        sys.exit(shard5(rank, world, ctx, m))

    if mode == "cbprod":
        # column-blocked layout of the distributed operator: ghost segments are column blocks of their own (VERDICT r3
        # item 6); the product must equal the single-GPU product of the whole matrix BIT FOR BIT (same additions in the
        # same order: rows are stored in global column order and the blocks follow it)
        n = m * m * m
        A = (ks.matrices.hashed_nonsymmetric_csr(n, seed=11) + sp.diags(np.linspace(1.0, 40.0, n))).tocsr()
        A.sort_indices()
        offs = ksd.partition_rows(n, world)
        r0, r1 = int(offs[rank]), int(offs[rank + 1])
        B = A[r0:r1]
        plan = ksd.build_halo_plan(B.indices.astype(np.int64), offs, rank, dist)
        op = ksd.dist_operator(api, ctx, B.indptr.astype(np.int64), B.data, plan, n)
        ws = api.ArnoldiWorkspace(r1 - r0, 5, np.float64, ctx=ctx, n_global=n, row_begin=r0)
        ctx1 = api.Context(device=local_rank)
        op1 = api.csr_operator(A, ctx=ctx1)
        ws1 = api.ArnoldiWorkspace(n, 5, np.float64, ctx=ctx1)
        ksd.ready_barrier(dist)
        want = os.environ.get("KS_EXPECT_LAYOUT", "csr-cb")
        lay = op.format["layout"]
        bad = 0
        import time
        tms = []
        for it in range(6):
            x = ks.matrices.uniform_hash(77 + it, np.arange(n)) - 0.5
            ws.set_col(0, x[r0:r1])
            ws1.set_col(0, x)
            ctx.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for c in range(4):
                ws.apply(op, c, c + 1)
            ctx.synchronize()
            tms.append((time.perf_counter() - t0) / 4)
            for c in range(4):
                ws1.apply(op1, c, c + 1)
            for c in (1, 4):
                bad += int(not np.array_equal(ws.col(c), ws1.col(c)[r0:r1]))
        # the product kernel alone (HIP events on the library's stream; the exchange kernels are not in this class)
        ctx.profile_enable(True)
        for c in range(4):
            ws.apply(op, c, c + 1)
        ctx.synchronize()
        pk = ctx.profile_get()["spmv"]
        ctx.profile_enable(False)
        kus = pk["ms"] / max(pk["count"], 1) * 1e3
        ok = bad == 0 and lay == want
        print(f"[rank {rank}] cbprod: product kernel {kus:.1f} us;", flush=True)
        print(f"[rank {rank}] cbprod: n={n} rows {r0}:{r1} ghosts={plan.nghost} layout={lay} (single GPU: {op1.format['layout']}) "
              f"{min(tms) * 1e6:.1f} us per product in a chain of 4 (all ranks on this device), "
              f"products differing from the single-GPU bits: {bad} -> {'OK' if ok else 'FAIL'}", flush=True)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)

    if mode in ("halo", "halohashed"):
        if mode == "halo":
            mx, my, mz = m, m + 1, m + 2 * world
            n = mx * my * mz
            offs = ksd.partition_rows(n, world, granule=mx * my)
            A = sp.csr_matrix(ks.matrices.laplace3d_csr(mx, my, mz, 0, n, index_dtype=np.int64)[::-1], shape=(n, n))
        else:
            n = m * m * m
            A = (ks.matrices.hashed_nonsymmetric_csr(n, seed=11) + sp.diags(np.linspace(1.0, 40.0, n))).tocsr()
            A.sort_indices()
            offs = ksd.partition_rows(n, world)
        r0, r1 = int(offs[rank]), int(offs[rank + 1])
        B = A[r0:r1]
        plan = ksd.build_halo_plan(B.indices.astype(np.int64), offs, rank, dist)
        op = ksd.dist_operator(api, ctx, B.indptr.astype(np.int64), B.data, plan, n)
        ws = api.ArnoldiWorkspace(r1 - r0, 5, np.float64, ctx=ctx, n_global=n, row_begin=r0)
        ksd.ready_barrier(dist)
        worst, bad = 0.0, 0
        for it in range(200):
            x = ks.matrices.uniform_hash(1000 + it, np.arange(n)) - 0.5
            chain = 1 + it % 4
            ws.set_col(0, x[r0:r1])
            for c in range(chain):  # columns 0 -> 1 -> 2 -> ...: no host synchronisation in between
                _lib.check(_lib.load().ks_apply(op._h, ws._h, c, c + 1))
            y = x
            for c in range(chain):
                y = A @ y
            got = ws.col(chain)
            err = float(np.abs(got - y[r0:r1]).max() / max(1e-300, np.abs(y).max()))
            worst = max(worst, err)
            bad += err > 1e-12
        ok = bad == 0
        print(f"[rank {rank}] {mode}: 200 chains of 1-4 products, rows {r0}:{r1}, neighbours={len(plan.neigh)}, layout={op.format['layout']}, "
              f"worst relative error {worst:.1e}, bad rounds {bad} -> {'OK' if ok else 'FAIL'}", flush=True)
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
        if mode == "outlier":
            A = ks.matrices.hashed_nonsymmetric_csr(n, seed=11, planted=[(30.0, 0.0), (25.0, 10.0), (-28.0, 0.0)])
        else:
            A = (ks.matrices.hashed_nonsymmetric_csr(n, seed=11) + sp.diags(diag)).tocsr()
        A.sort_indices()
        offs = ksd.partition_rows(n, world)
        r0, r1 = int(offs[rank]), int(offs[rank + 1])
        B = A[r0:r1]
        ip, ix, dv = B.indptr.astype(np.int64), B.indices.astype(np.int64), B.data
        kw = dict(nev=5, which="LR", tol=1e-10, mindim=10, maxdim=25, restarts=300)
        if mode == "wide":
            kw = dict(nev=12, which="LR", tol=1e-10, mindim=30, maxdim=60, restarts=300)
        if mode == "eager":
            kw = dict(nev=12, which="LR", tol=1e-10, mindim=35, maxdim=70, restarts=300)
        if mode == "outlier":
            kw = dict(nev=5, which="LM", tol=1e-10, mindim=10, maxdim=30, restarts=300)
        full = lambda: A  # noqa: E731
    dtype = np.complex128 if mode == "complex" else np.float64

    plan = ksd.build_halo_plan(ix, offs, rank, dist)
    op = ksd.dist_operator(api, ctx, ip, dv, plan, n)
    ws = api.ArnoldiWorkspace(r1 - r0, kw["maxdim"], dtype, ctx=ctx, n_global=n, row_begin=r0)
    ws._v1 = ks.matrices.start_vector(r1 - r0, row_begin=r0).astype(dtype)
    setup_skew(rank)
    ksd.ready_barrier(dist)
    F, hist = ks.partialschur_(op, ws, **kw)
    res, orth = ws.residual_norms(op, F.nconverged)
    # ||A Q - Q R||_F over nconverged columns; the solver's criterion is per vector, tol * |lambda| (src/run.jl:330)
    ok = bool(hist.converged and res < 10 * kw["tol"] * np.abs(F.eigenvalues).max() * max(1, F.nconverged) and orth < 1e-12)
    msg = f"{hist} resid={res:.2e} orth={orth:.2e} neighbours={len(plan.neigh)} rows {r0}:{r1}"
    if os.environ.get("KS_CHECK_BLOCKS") == "1":  # the s-step expansion must really have run in blocks on every rank
        info = ws.sstep_info
        msg += f" sstep={info['s']} blocks{'>0' if info['blocks'] > 0 else '=0'} ({info['blocks']}, abandoned {info['abandoned']})"
        if mode == "outlier":
            msg += f" deflated{'>0' if info['deflated_blocks'] > 0 else '=0'} ({info['deflated_blocks']}, columns {info['deflated_columns']})"
            ok = ok and info["deflated_blocks"] > 0 and info["abandoned"] == 0
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
