"""Synthetic operands of the benchmark / parity configurations, generated
directly as CSR arrays with numpy (no Kronecker products, so n = 10^7 takes seconds)."""
from __future__ import annotations

import numpy as np


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform_hash(seed: int, idx) -> np.ndarray:
    """u[i] = (splitmix64(seed xor i) >> 11) * 2^-53 -- the RNG shared with the HIP kernels."""
    h = splitmix64(np.uint64(seed & ((1 << 64) - 1)) ^ np.asarray(idx, dtype=np.uint64))
    return (h >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def start_vector(n: int, seed: int = 20240917, row_begin: int = 0) -> np.ndarray:
    return uniform_hash(seed, np.arange(row_begin, row_begin + n, dtype=np.uint64))


def laplace1d_csr(n: int):
    """spdiagm(-1 => -1, 0 => 2, 1 => -1) (readme.md:28-32) as (indptr, indices, data)."""
    return _stencil_rows((n, 1, 1), 0, n, diag=2.0)


def laplace3d_csr(mx: int, my: int, mz: int, row_begin: int = 0, row_end: int | None = None, index_dtype=np.int32):
    """Rows [row_begin, row_end) of the 7-point Dirichlet Laplacian on an mx x my x mz grid
    (row i = x + mx*(y + my*z); diagonal 6, off-diagonals -1), GLOBAL column indices, sorted."""
    n = mx * my * mz
    if row_end is None:
        row_end = n
    return _stencil_rows((mx, my, mz), row_begin, row_end, diag=6.0, index_dtype=index_dtype)


def laplace3d_csr_chunked(mx: int, my: int, mz: int, index_dtype=np.int32, chunk_rows: int = 1 << 22):
    """The whole 7-point Laplacian assembled chunk by chunk into preallocated arrays: same result as
    `laplace3d_csr(mx, my, mz)` with O(chunk) temporaries (n = 10^8 needs ~9 GB instead of ~25 GB of host memory)."""
    n = mx * my * mz
    nnz = 7 * n - 2 * (mx * my + mx * mz + my * mz)
    indptr = np.empty(n + 1, dtype=np.int64)
    indices = np.empty(nnz, dtype=index_dtype)
    data = np.empty(nnz, dtype=np.float64)
    indptr[0] = 0
    pos = 0
    for r0 in range(0, n, chunk_rows):
        r1 = min(n, r0 + chunk_rows)
        ip, ix, dv = _stencil_rows((mx, my, mz), r0, r1, diag=6.0, index_dtype=index_dtype)
        cnt = int(ip[-1])
        indptr[r0 + 1 : r1 + 1] = ip[1:] + pos
        indices[pos : pos + cnt] = ix
        data[pos : pos + cnt] = dv
        pos += cnt
    assert pos == nnz
    return indptr, indices, data


def _stencil_rows(dims, r0, r1, diag, index_dtype=np.int32):
    mx, my, mz = dims
    rows = np.arange(r0, r1, dtype=np.int64)
    x = rows % mx
    y = (rows // mx) % my
    z = rows // (mx * my)
    nloc = rows.shape[0]
    offs = [(-mx * my, z > 0), (-mx, y > 0), (-1, x > 0), (0, np.ones(nloc, bool)), (1, x < mx - 1), (mx, y < my - 1), (mx * my, z < mz - 1)]
    if my == 1 and mz == 1:
        offs = [offs[2], offs[3], offs[4]]
    mask = np.stack([m for _, m in offs], axis=1)
    cand = np.stack([rows + o for o, _ in offs], axis=1)
    vals = np.where(np.array([o for o, _ in offs]) == 0, diag, -1.0)[None, :].repeat(nloc, 0)
    indptr = np.zeros(nloc + 1, dtype=np.int64)
    np.cumsum(mask.sum(1), out=indptr[1:])
    return indptr, cand[mask].astype(index_dtype), vals[mask].astype(np.float64)


def laplace3d_eigs(mx: int, my: int, mz: int, k: int | None = None):
    ex = 2 - 2 * np.cos(np.arange(1, mx + 1) * np.pi / (mx + 1))
    ey = 2 - 2 * np.cos(np.arange(1, my + 1) * np.pi / (my + 1))
    ez = 2 - 2 * np.cos(np.arange(1, mz + 1) * np.pi / (mz + 1))
    ev = np.sort((ex[:, None, None] + ey[None, :, None] + ez[None, None, :]).ravel())
    return ev if k is None else ev[:k]


def to_scipy(indptr, indices, data, ncols):
    import scipy.sparse as sp

    return sp.csr_matrix((data, indices, indptr), shape=(len(indptr) - 1, ncols))


def hashed_nonsymmetric_csr(n: int, seed: int = 7, planted=None):
    """Portable stand-in for sprand(n, n, 5/n): row i has 1 + (h mod 9) entries at hashed columns,
    values uniform [0,1); duplicates summed.  `planted`: list of (a, b) -> the leading rows are replaced by 2x2 blocks
    [a b; -b a] on the diagonal (b == 0: one row with the real entry a): exact, separated eigenvalues a +- i b of the
    matrix (the planted rows have no other entries), as oracle.matrices.hashed_nonsymmetric plants them.  Returns a scipy CSR matrix."""
    import scipy.sparse as sp

    idx = np.arange(n, dtype=np.uint64)
    deg = 1 + (uniform_hash(seed, idx * np.uint64(64)) * 9).astype(np.int64)
    rows, cols, vals = [], [], []
    for t in range(9):
        m = deg > t
        r = idx[m]
        c = np.minimum((uniform_hash(seed + 1, r * np.uint64(64) + np.uint64(t + 1)) * n).astype(np.int64), n - 1)
        rows.append(r.astype(np.int64))
        cols.append(c)
        vals.append(uniform_hash(seed + 2, r * np.uint64(64) + np.uint64(t + 1)))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    if planted:
        pr, pc, pv, p = [], [], [], 0
        for a, b in planted:
            if b == 0:
                pr += [p]; pc += [p]; pv += [a]
                p += 1
            else:
                pr += [p, p, p + 1, p + 1]; pc += [p, p + 1, p, p + 1]; pv += [a, b, -b, a]
                p += 2
        keep = rows >= p
        rows = np.concatenate([rows[keep], np.array(pr, dtype=np.int64)])
        cols = np.concatenate([cols[keep], np.array(pc, dtype=np.int64)])
        vals = np.concatenate([vals[keep], np.array(pv, dtype=np.float64)])
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A
