"""ORACLE (test infrastructure): independent constructions of the synthetic operands of
SURVEY.md section 8d, built with scipy.sparse (Kronecker sums), used to cross-check the
product's own generators and to feed the oracle solver."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .arnoldi import uniform_hash


def laplace1d(n: int):
    """spdiagm(-1 => -1, 0 => 2, 1 => -1)  (readme.md:28-32)."""
    return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")


def laplace3d(mx: int, my: int, mz: int):
    """7-point Dirichlet Laplacian, row index i = x + mx*(y + my*z), diagonal 6, off-diagonals -1."""
    Ix, Iy, Iz = sp.identity(mx), sp.identity(my), sp.identity(mz)
    A = (
        sp.kron(Iz, sp.kron(Iy, laplace1d(mx)))
        + sp.kron(Iz, sp.kron(laplace1d(my), Ix))
        + sp.kron(laplace1d(mz), sp.kron(Iy, Ix))
    ).tocsr()
    A.sort_indices()
    return A


def laplace3d_eigs(mx: int, my: int, mz: int):
    """All eigenvalues, ascending: sum_d (2 - 2 cos(i_d pi / (m_d + 1)))."""
    ex = 2 - 2 * np.cos(np.arange(1, mx + 1) * np.pi / (mx + 1))
    ey = 2 - 2 * np.cos(np.arange(1, my + 1) * np.pi / (my + 1))
    ez = 2 - 2 * np.cos(np.arange(1, mz + 1) * np.pi / (mz + 1))
    return np.sort((ex[:, None, None] + ey[None, :, None] + ez[None, None, :]).ravel())


def hashed_nonsymmetric(n: int, seed: int = 7, planted=None):
    """Language-portable stand-in for `sprand(n, n, 5/n)` (test/expansion.jl:16):
    row i has d_i = 1 + (h mod 9) entries at hashed columns with uniform [0,1) values;
    duplicates are summed.  `planted`: list of (a, b) -> 2x2 blocks [a b; -b a] on the
    leading diagonal (b == 0 -> a real 1x1 spike), scaled-down random part elsewhere."""
    rows, cols, vals = [], [], []
    idx = np.arange(n, dtype=np.uint64)
    deg = 1 + (uniform_hash(seed, idx * np.uint64(64)) * 9).astype(np.int64)
    for t in range(9):
        m = deg > t
        r = idx[m]
        c = (uniform_hash(seed + 1, r * np.uint64(64) + np.uint64(t + 1)) * n).astype(np.int64)
        v = uniform_hash(seed + 2, r * np.uint64(64) + np.uint64(t + 1))
        rows.append(r.astype(np.int64))
        cols.append(np.minimum(c, n - 1))
        vals.append(v)
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    A.sum_duplicates()
    if planted:
        A = A.tolil()
        p = 0
        for a, b in planted:
            if b == 0:
                A[p, :] = 0
                A[p, p] = a
                p += 1
            else:
                A[p, :] = 0
                A[p + 1, :] = 0
                A[p, p] = a
                A[p, p + 1] = b
                A[p + 1, p] = -b
                A[p + 1, p + 1] = a
                p += 2
        A = A.tocsr()
    A.sort_indices()
    return A
