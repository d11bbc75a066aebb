"""Worker for tests/test_dist_gloo.py: run under torch.distributed.run with the gloo backend (CPU).

Checks, on world_size ranks:
  1. partition + halo plan (arnoldimethod.jl_amd/dist.py): executing the plan with point-to-point
     messages and multiplying with local-extended column indices reproduces the global SpMV;
  2. the sharded Arnoldi expansion (row-local kernels + all-reduce of the DGKS coefficients and norms,
     every rank taking the same branch) reproduces the single-process oracle's H to 1e-12 and yields
     the same Ritz values;
  3. the counter-based start vector / RNG is partition independent.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from __graft_entry__ import import_package  # noqa: E402
from oracle import arnoldi as oa  # noqa: E402
from oracle.matrices import hashed_nonsymmetric, laplace3d  # noqa: E402

pkg = import_package()
from arnoldimethod_jl_amd import dist as ksd  # noqa: E402

ETA = np.sqrt(2) / 2


def allsum(x):
    t = torch.from_numpy(np.atleast_1d(np.asarray(x, dtype=np.float64)).copy())
    dist.all_reduce(t)
    return t.numpy()


def local_block(A, r0, r1):
    B = A[r0:r1].tocsr()
    B.sort_indices()
    return B.indptr.astype(np.int64), B.indices.astype(np.int64), B.data.astype(np.float64)


def dist_spmv(plan, ip, dv, x_local, rank):
    ghost = ksd.halo_exchange_host(plan, x_local, dist, rank)
    xe = np.concatenate([x_local, ghost])
    import scipy.sparse as sp

    M = sp.csr_matrix((dv, plan.colidx_local, ip), shape=(plan.n_local, plan.n_local + plan.nghost))
    return M @ xe


def check_matrix(A, offs, rank, world, tag):
    n = A.shape[0]
    r0, r1 = int(offs[rank]), int(offs[rank + 1])
    ip, ix, dv = local_block(A, r0, r1)
    plan = ksd.build_halo_plan(ix, offs, rank, dist)
    # plan consistency: what I send to q is what q expects from me
    counts = [None] * world
    dist.all_gather_object(counts, {int(q): (int(plan.send_ptr[p + 1] - plan.send_ptr[p]), int(plan.recv_cnt[p])) for p, q in enumerate(plan.neigh)})
    for p, q in enumerate(plan.neigh):
        assert counts[q][rank][1] == plan.send_ptr[p + 1] - plan.send_ptr[p], (tag, "send/recv mismatch")
        assert counts[q][rank][0] == plan.recv_cnt[p]
    assert (np.diff(plan.ghost_global) > 0).all()
    # 1. SpMV
    x = oa.uniform_hash(99, np.arange(n)) - 0.5
    y = dist_spmv(plan, ip, dv, x[r0:r1], rank)
    np.testing.assert_allclose(y, (A @ x)[r0:r1], rtol=1e-14, atol=1e-14)
    # 2. sharded Arnoldi with DGKS, decisions on globally reduced norms
    m = 12
    v1 = pkg.matrices.start_vector(n)               # global (oracle) ...
    v1_loc = pkg.matrices.start_vector(r1 - r0, row_begin=r0)  # ... == the slice generated locally
    assert (v1[r0:r1] == v1_loc).all()
    ows = oa.ArnoldiWorkspace.from_vector(v1, m)
    oa.reinitialize(ows, 0, lambda v: v.__setitem__(slice(None), v1))
    st = {}
    oa.iterate_arnoldi(A, ows, 1, m, st)
    V = np.zeros((r1 - r0, m + 1))
    H = np.zeros((m + 1, m))
    V[:, 0] = v1_loc / np.sqrt(allsum(v1_loc @ v1_loc)[0])
    nre = 0
    for j in range(1, m + 1):
        w = dist_spmv(plan, ip, dv, V[:, j - 1], rank)
        red = allsum(np.concatenate([V[:, :j].T @ w, [w @ w]]))      # one all-reduce of j+1 doubles
        h, rnorm = red[:j], np.sqrt(red[j])
        w = w - V[:, :j] @ h
        wnorm = np.sqrt(allsum(w @ w)[0])                             # one all-reduce of 1 double
        if wnorm < ETA * rnorm:
            rnorm = wnorm
            c = allsum(V[:, :j].T @ w)
            w = w - V[:, :j] @ c
            h = h + c
            wnorm = np.sqrt(allsum(w @ w)[0])
            nre += 1
        assert not wnorm <= ETA * rnorm
        H[:j, j - 1] = h
        H[j, j - 1] = wnorm
        V[:, j] = w / wnorm
    assert nre == st.get("reorth", 0), (tag, nre, st)
    np.testing.assert_allclose(H, ows.H, atol=1e-12)
    np.testing.assert_allclose(V, ows.V[r0:r1], atol=1e-10)
    # 3. the S-STEP (block) expansion, sharded: steps 5..12 in blocks of 4 on top of the first four columns.  The n-sized
    # work is row-local; per block TWO all-reduces ([S Z]^H Z, then [S Qt]^H Qt) and a halo exchange per product; the small
    # algebra (tests/sstep_model.py = csrc/ks_block_kernels.hpp: k_fin_blk) runs replicated on every rank from identical sums.
    import sstep_model as sm

    def inner(X, Y):
        loc = X.conj().T @ Y
        return allsum(loc.ravel()).reshape(loc.shape)

    stf = sm.Factored(r1 - r0, m, np.float64)
    stf.S[:, :5] = V[:, :5]
    stf.H[:, :4] = H[:, :4]
    shifts = sm.newton_shifts(np.linalg.eigvals(ows.H[:m, :m]), 4, True)
    sm.expand_block2(A, stf, 5, m, shifts, 4, {}, scale=1.0 / 8.0, inner=inner, apply=lambda x: dist_spmv(plan, ip, dv, x, rank))
    Vt = stf.true_basis(m + 1)
    np.testing.assert_allclose(stf.H, ows.H, atol=1e-11)
    np.testing.assert_allclose(Vt, ows.V[r0:r1], atol=1e-9)
    allH = [None] * world
    dist.all_gather_object(allH, stf.H.tobytes())
    assert all(h == allH[0] for h in allH), (tag, "ranks disagree on H: the replicated algebra must see identical sums")
    # 4. the same blocks with IN-CHAIN DEFLATION against the first two columns (csrc/ks_block.hpp: k_defl_dots -> k_defl_reduce ->
    # all-reduce -> k_defl_apply): the identity  A z_{i-1} = z_i / sigma_i + theta_i z_{i-1} + U c_i / sigma_i  holds for ANY columns
    # U of the basis, so the same H and the same basis must come out; per product one more all-reduce (of ndefl doubles)
    std = sm.Factored(r1 - r0, m, np.float64)
    std.S[:, :5] = V[:, :5]
    std.H[:, :4] = H[:, :4]
    sm.expand_block2(A, std, 5, m, shifts, 4, {}, scale=1.0 / 8.0, inner=inner, apply=lambda x: dist_spmv(plan, ip, dv, x, rank), ndefl=2)
    np.testing.assert_allclose(std.H, ows.H, atol=1e-11)
    np.testing.assert_allclose(std.true_basis(m + 1), ows.V[r0:r1], atol=1e-9)
    dist.all_gather_object(allH, std.H.tobytes())
    assert all(h == allH[0] for h in allH), (tag, "ranks disagree on H with the deflated chain")
    return True


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # slab-partitioned 3-D Laplacian (whole planes per rank, like bench.py --gpus N)
    mx, my, mz = 6, 7, 8
    A = laplace3d(mx, my, mz)
    offs = ksd.partition_rows(A.shape[0], world, granule=mx * my)
    assert offs[-1] == A.shape[0] and (np.diff(offs) % (mx * my) == 0).all()
    check_matrix(A, offs, rank, world, "laplace3d")
    # product generator: the slab each rank builds stacks to the same matrix
    ip, ix, dv = pkg.matrices.laplace3d_csr(mx, my, mz, int(offs[rank]), int(offs[rank + 1]), index_dtype=np.int64)
    ip2, ix2, dv2 = local_block(A, int(offs[rank]), int(offs[rank + 1]))
    assert (ip == ip2).all() and (ix == ix2).all() and (dv == dv2).all()
    plan = ksd.build_halo_plan(ix, offs, rank, dist)
    # each interior slab boundary exchanges exactly one plane
    for p, q in enumerate(plan.neigh):
        assert abs(int(q) - rank) == 1 and plan.recv_cnt[p] == mx * my
    # irregular, nonsymmetric pattern with an uneven partition (general gather lists, one-sided needs)
    B = (hashed_nonsymmetric(500, seed=5) + 3.0 * __import__("scipy.sparse").sparse.identity(500)).tocsr()
    offs2 = np.array([0, 137, 500]) if world == 2 else ksd.partition_rows(500, world)
    check_matrix(B, offs2, rank, world, "hashed")
    dist.barrier()
    if rank == 0:
        print("DIST_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
