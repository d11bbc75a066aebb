"""Row-partitioned multi-GPU layer (one process per GPU).

Rows of A and of the Krylov basis V are split into contiguous blocks, one per rank.  The only
exchanges on the hot path are
  * the ghost entries of x before every SpMV (neighbour send/recv over xGMI, executed by the library
    from the plan built here), and
  * the tiny all-reduces of the Gram-Schmidt coefficients / norms (inside the library, RCCL).
H, Q and all Schur logic are replicated: every rank sees bit-identical reduced values and takes the
same branches (src/expansion.jl:91,99).

This module contains no GPU code: it computes the partition and the halo plan with numpy and a
`torch.distributed` process group of ANY backend (gloo on CPU in the tests, nccl == RCCL on the GPU
box), and hands plain arrays to `ks_operator_csr_dist` (include/kschur.h).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check


def partition_rows(n: int, world: int, granule: int = 1) -> np.ndarray:
    """Row offsets (world+1) of a balanced contiguous partition whose cuts are multiples of `granule`
    (granule = mx*my keeps whole grid planes of a 3-D stencil on one rank)."""
    units = n // granule
    assert units * granule == n, "n must be a multiple of the granule"
    base, rem = divmod(units, world)
    sizes = np.array([base + (1 if r < rem else 0) for r in range(world)], dtype=np.int64) * granule
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


@dataclass
class HaloPlan:
    """What rank `rank` must exchange before y = A x (see ks_operator_csr_dist)."""

    n_local: int
    nghost: int
    neigh: np.ndarray      # int32  [nneigh]   neighbour ranks, ascending
    send_ptr: np.ndarray   # int64  [nneigh+1] offsets into send_idx
    send_idx: np.ndarray   # int32  [nsend]    LOCAL row indices to pack, per neighbour
    recv_cnt: np.ndarray   # int64  [nneigh]   ghost values received from each neighbour (consecutive slots)
    ghost_global: np.ndarray  # int64 [nghost] global column of every ghost slot (tests / debugging)
    colidx_local: np.ndarray  # int32 [nnz]    local-extended column indices


def build_halo_plan(indices_global: np.ndarray, row_offsets: np.ndarray, rank: int, dist=None, exchange=None) -> HaloPlan:
    """`indices_global`: global column indices of this rank's CSR rows.  `dist`: torch.distributed (any
    backend) -- or pass `exchange(list_of_requests) -> list over ranks` for tests without a process group."""
    world = len(row_offsets) - 1
    r0, r1 = int(row_offsets[rank]), int(row_offsets[rank + 1])
    n_local = r1 - r0
    idx = np.asarray(indices_global, dtype=np.int64)
    owned = (idx >= r0) & (idx < r1)
    ghosts = np.unique(idx[~owned])                                     # sorted global columns
    owner = np.searchsorted(row_offsets, ghosts, side="right") - 1      # ascending with the column
    # ghost slot g <-> ghosts[g]: ordered by owner rank, then by global index (== plain sort)
    colidx_local = np.empty(idx.shape, dtype=np.int32)
    colidx_local[owned] = (idx[owned] - r0).astype(np.int32)
    colidx_local[~owned] = (n_local + np.searchsorted(ghosts, idx[~owned])).astype(np.int32)
    # what I need from every other rank (global indices, ascending)
    need = [ghosts[owner == q] for q in range(world)]
    if exchange is not None:
        all_need = exchange(need)
    elif world == 1:
        all_need = [need]
    else:
        all_need = [None] * world
        dist.all_gather_object(all_need, [a.tolist() for a in need])
        all_need = [[np.asarray(x, dtype=np.int64) for x in per_rank] for per_rank in all_need]
    # all_need[q][p] = what rank q needs from rank p  ->  what I (rank) must send to q
    send_to = {q: np.asarray(all_need[q][rank], dtype=np.int64) for q in range(world) if q != rank and len(all_need[q][rank])}
    recv_from = {q: need[q] for q in range(world) if q != rank and len(need[q])}
    neigh = sorted(set(send_to) | set(recv_from))
    send_ptr = [0]
    send_idx = []
    recv_cnt = []
    for q in neigh:
        s = send_to.get(q, np.zeros(0, dtype=np.int64))
        assert ((s >= r0) & (s < r1)).all(), "peer requested rows this rank does not own"
        send_idx.append((s - r0).astype(np.int32))
        send_ptr.append(send_ptr[-1] + len(s))
        recv_cnt.append(len(recv_from.get(q, ())))
    return HaloPlan(
        n_local=n_local,
        nghost=int(len(ghosts)),
        neigh=np.asarray(neigh, dtype=np.int32),
        send_ptr=np.asarray(send_ptr, dtype=np.int64),
        send_idx=(np.concatenate(send_idx) if send_idx else np.zeros(0, dtype=np.int32)).astype(np.int32),
        recv_cnt=np.asarray(recv_cnt, dtype=np.int64),
        ghost_global=ghosts,
        colidx_local=colidx_local,
    )


def dist_operator(api, ctx, indptr, data, plan: HaloPlan, n_global: int):
    """Upload this rank's CSR block + halo plan (ks_operator_csr_dist)."""
    L = _lib.load()
    dt = np.complex128 if np.asarray(data).dtype.kind == "c" else np.float64
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    col = np.ascontiguousarray(plan.colidx_local, dtype=np.int32)
    val = np.ascontiguousarray(data, dtype=dt)
    neigh = np.ascontiguousarray(plan.neigh, dtype=np.int32)
    sp_ = np.ascontiguousarray(plan.send_ptr, dtype=np.int64)
    si = np.ascontiguousarray(plan.send_idx, dtype=np.int32)
    rc = np.ascontiguousarray(plan.recv_cnt, dtype=np.int64)
    h = C.c_void_p()
    check(
        L.ks_operator_csr_dist(
            ctx._h, plan.n_local, plan.nghost, len(val), indptr.ctypes.data, col.ctypes.data, val.ctypes.data,
            _lib.KS_C64 if dt == np.complex128 else _lib.KS_F64, len(neigh), neigh.ctypes.data, sp_.ctypes.data,
            si.ctypes.data, rc.ctypes.data, C.byref(h),
        )
    )
    return api.Operator(ctx, h, (n_global, n_global), dt)


def make_context(api, dist, local_rank: int, transport: str | None = None):
    """One library context per rank.

    transport "rccl" (default): rank 0's RCCL unique id is broadcast through the process group.
    transport "host": the host-staged transport on the same process group (debugging / RCCL-free fabrics).
    transport "p2p": no RCCL at all -- every rank allocates its shared region, the 64-byte IPC handles are
    all-gathered through the process group (any backend) and mapped (csrc/ks_p2p.hpp).  KS_TRANSPORT
    selects the default."""
    import os

    rank, world = dist.get_rank(), dist.get_world_size()
    transport = transport or os.environ.get("KS_TRANSPORT", "rccl")
    # the host driver only supports dmabuf IPC: without this RCCL and the peer-to-peer regions fail with
    # `hipIpcGetMemHandle: invalid argument` (read when the HSA runtime starts, i.e. at the first device call)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if transport == "p2p" and world > 1:
        # Ranks that SHARE a device (several processes on one GPU: how the multi-rank path is tested on a one-GPU box) must
        # not use the exchange folded into the SpMV launch: there the workgroups of a slab's boundary planes WAIT inside the
        # kernel for the neighbours' entries (1 700 workgroups per rank at 464^2 rows per plane), and on a shared device they
        # hold the very CU slots the neighbours' pushing workgroups need -- a deadlock that only the wall-clock budget ends
        # (CommTimeout; seen with >= 4 ranks from 232^3 on, never with one rank per GPU, where a rank's own pushers are
        # dispatched first and its waiters only ever wait for ANOTHER device).  The push kernel in front of the SpMV
        # (KS_HALO_FUSED=0) has no such coupling.  An explicit KS_HALO_FUSED wins.
        import socket

        where = [None] * world
        dist.all_gather_object(where, (socket.gethostname(), int(local_rank)))
        if len(set(where)) < world:
            os.environ.setdefault("KS_HALO_FUSED", "0")
    if transport == "p2p":
        ctx = api.Context(local_rank, rank, world, p2p=True)
        if world > 1:
            handles = [None] * world
            dist.all_gather_object(handles, ctx.p2p_handle())
            ctx.p2p_attach(handles)
            dist.barrier()
        return ctx
    if transport == "host":
        return api.Context(local_rank, rank, world, hostcomm=host_transport(dist))
    box = [api.Context.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return api.Context(local_rank, rank, world, box[0])


def ready_barrier(dist):
    """Control-plane rendezvous between SET-UP and the first exchange.  Every wait inside the library is bounded by wall
    clock (KS_P2P_TIMEOUT_S, 30 s; RCCL has its own watchdog) and the clock starts when the first exchange kernel of the
    FASTEST rank starts -- while a slower rank may still be assembling its row block on the host (8 processes on a 16-CPU
    quota: tens of seconds at 464^3 / 8; round 3's committed config-5 evidence died exactly there with CommTimeout).
    Call after the operator and the workspace exist on every rank and before the first verb that exchanges."""
    import os

    if dist.get_world_size() > 1 and os.environ.get("KS_NO_READY_BARRIER") != "1":   # (the switch exists for A/B experiments only)
        dist.barrier()


def host_transport(dist):
    """The two callables of the host-staged transport (ks_ctx_create_hostcomm) on a torch.distributed process group
    of any backend with CPU tensors (gloo): sum in place / grouped neighbour exchange.  gloo's all-reduce is
    deterministic and delivers the same bits to every rank, which the replicated DGKS decisions need."""
    import torch

    def allreduce(buf):
        t = torch.from_numpy(buf)
        dist.all_reduce(t)

    def exchange(peers, sendbufs, recvbufs):
        reqs = []
        for q, sb, rb in zip(peers, sendbufs, recvbufs):
            if len(sb):
                reqs.append(dist.isend(torch.from_numpy(sb), q))
            if len(rb):
                reqs.append(dist.irecv(torch.from_numpy(rb), q))
        for r in reqs:
            r.wait()

    return allreduce, exchange


def setup_laplace3d(pkg, dist, m: int, maxdim: int, local_rank: int, transport: str | None = None):
    """bench.py N > 1: slab partition of the m^3 Laplacian along z, one slab of whole planes per rank."""
    from . import api

    rank, world = dist.get_rank(), dist.get_world_size()
    n = m ** 3
    offs = partition_rows(n, world, granule=m * m)
    r0, r1 = int(offs[rank]), int(offs[rank + 1])
    ip, ix, dv = pkg.matrices.laplace3d_csr(m, m, m, r0, r1, index_dtype=np.int64)
    plan = build_halo_plan(ix, offs, rank, dist)
    ctx = make_context(api, dist, local_rank, transport)
    op = dist_operator(api, ctx, ip, dv, plan, n)
    ws = api.ArnoldiWorkspace(r1 - r0, maxdim, np.float64, ctx=ctx, n_global=n, row_begin=r0)
    v1 = pkg.matrices.start_vector(r1 - r0, row_begin=r0)
    nnz = 7 * n - 6 * m * m  # 7-point stencil with Dirichlet faces
    ready_barrier(dist)  # nobody starts exchanging before the slowest rank finished its set-up
    return ctx, op, ws, v1, nnz


# ------------------------------------------------------------------ host reference of the exchange
def halo_exchange_host(plan: HaloPlan, x_local: np.ndarray, dist, rank: int) -> np.ndarray:
    """CPU execution of the plan with torch.distributed point-to-point ops (used by the gloo tests to
    prove the plan is right: the library does exactly this with ncclSend/ncclRecv on device buffers)."""
    import torch

    ghost = np.zeros(plan.nghost, dtype=x_local.dtype)
    reqs = []
    recv_bufs = []
    goff = 0
    for p, q in enumerate(plan.neigh):
        s = plan.send_idx[plan.send_ptr[p] : plan.send_ptr[p + 1]]
        if len(s):
            t = torch.from_numpy(np.ascontiguousarray(x_local[s]))
            reqs.append(dist.isend(t, int(q)))
        cnt = int(plan.recv_cnt[p])
        if cnt:
            buf = torch.empty(cnt, dtype=torch.from_numpy(x_local[:1].copy()).dtype)
            reqs.append(dist.irecv(buf, int(q)))
            recv_bufs.append((goff, cnt, buf))
        goff += cnt
    for r in reqs:
        r.wait()
    for off, cnt, buf in recv_bufs:
        ghost[off : off + cnt] = buf.numpy()
    return ghost
