"""`-m "not gpu"`: a numpy MODEL of the implicit second DGKS pass the HIP expansion uses by default (ks_kernels.hpp,
k_fin_dots_t / k_fin_mid_t), checked against the oracle's reference-order expansion.

The reference applies the second projection to the vector (src/expansion.jl:93-94: c = V'v, v -= V c).  The device path never
does: the column stays in memory as the FIRST projection w' and the basis is carried as V_true = S T with a small upper
triangular T.  Everything the expansion needs from V_true follows from what the two streaming kernels deliver for S:

    y' = A S[:, j-1]                  A v_true = (y' - V_true g) / beta,  g = H[0:j, 0:j-1] c   (linearity + Arnoldi relation)
    s  = S^H y'                       t = T^H s;  h = (t - g) / beta;  ||A v_true||^2 = (|y'|^2 - 2 Re g^H t + |g|^2) / beta^2
    w' = y'/beta - S (T t / beta)     (= (I - V V^H) y' / beta)
    c_raw = S^H w', ||w'||^2          c = T^H c_raw;  DGKS test;  h += c;  beta_new = sqrt(||w'||^2 - ||c||^2);
                                      T[:, j] = ( -(T c) / beta_new ; 1 / beta_new )
    restart:  V_true Q = S (T Q), residual column = S T[:, m]; afterwards T = I.

This file is the executable statement of that algebra (what the device kernels are tested against on the GPU is the oracle
itself; here the ALGORITHM is pinned on the CPU): identical restart trails and matrix-vector counts, residuals and
orthogonality at the oracle's level, Ritz values to 1e-12 -- including the delicate case (low-rank operator, near-breakdown)
that dictates the form of the DGKS test."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import arnoldi as oa
from oracle import smalldense as sd
from oracle.matrices import laplace3d

ETA = oa.ETA
EPS = np.finfo(np.float64).eps


class Factored:
    def __init__(self, n, maxdim, dtype):
        self.S = np.zeros((n, maxdim + 1), dtype=dtype, order="F")  # stored columns
        self.T = np.eye(maxdim + 1, dtype=dtype)                    # V_true = S T
        self.H = np.zeros((maxdim + 1, maxdim), dtype=dtype, order="F")
        self.c = {}                                                  # second-pass coefficients of the factored columns

    def true_basis(self, ncols):
        return self.S[:, :ncols] @ self.T[:ncols, :ncols]


def expand(A, st, frm, to, stats):
    S, T, H = st.S, st.T, st.H
    n = S.shape[0]
    for j in range(frm, to + 1):
        yp = A @ S[:, j - 1]                                 # the operator is applied to the STORED column
        c = st.c.get(j - 1)
        beta = 1.0 / T[j - 1, j - 1].real
        g = np.zeros(j, dtype=H.dtype)
        if c is not None and len(c):
            g[: len(c) + 1] = H[: len(c) + 1, : len(c)] @ c   # g = H[0:j, 0:j-1] c
        s = S[:, :j].conj().T @ yp                           # pass 1 over the basis (k_dots)
        yy = np.vdot(yp, yp).real
        t = T[:j, :j].conj().T @ s
        h = (t - g) / beta
        rnorm = np.sqrt(max(0.0, yy - 2 * np.vdot(g, t).real + np.vdot(g, g).real)) / beta
        rproj = np.sqrt(yy) / beta                           # norm of what the projection really works on
        wp = yp / beta - S[:, :j] @ (T[:j, :j] @ t / beta)   # pass 2 (k_axpy_dots_cs) ...
        S[:, j] = wp
        craw = S[:, :j].conj().T @ wp                        # ... which also delivers these
        wnorm = float(np.linalg.norm(wp))
        ctrue = T[:j, :j].conj().T @ craw
        if wnorm < ETA * max(rnorm, rproj):                  # src/expansion.jl:91 (see k_fin_mid_t for the max)
            stats["reorth"] = stats.get("reorth", 0) + 1
            h = h + ctrue
            cc = ctrue
            bnew = np.sqrt(max(wnorm ** 2 - np.vdot(cc, cc).real, 0.0))
            rnorm_p = wnorm
        else:
            cc = np.zeros(j, dtype=H.dtype)
            bnew = wnorm
            rnorm_p = rnorm
        H[:j, j - 1] = h
        stats["steps"] = stats.get("steps", 0) + 1
        if bnew <= ETA * rnorm_p:                            # :99-102, then reinitialize! (:127-129)
            H[j, j - 1] = 0.0
            S[:, :j] = st.true_basis(j)
            T[:, :] = np.eye(T.shape[0], dtype=H.dtype)
            st.c = {}
            if j != n:
                v = oa.uniform_hash(1000 + j, np.arange(n)).astype(H.dtype)
                for _ in range(2):
                    v -= S[:, :j] @ (S[:, :j].conj().T @ v)
                S[:, j] = v / np.linalg.norm(v)
                stats["breakdowns"] = stats.get("breakdowns", 0) + 1
            continue
        H[j, j - 1] = bnew
        T[:, j] = 0
        T[:j, j] = -(T[:j, :j] @ cc) / bnew
        T[j, j] = 1.0 / bnew
        st.c[j] = cc


def solve(A, v1, nev, which, tol, mindim, maxdim, restarts, dtype):
    """oracle/arnoldi.py:_partialschur with the factored expansion and the T-folded rotation."""
    n = A.shape[0]
    st = Factored(n, maxdim, dtype)
    st.S[:, 0] = v1 / np.linalg.norm(v1)
    H, Q = st.H, np.zeros((maxdim, maxdim), dtype=dtype, order="F")
    real = np.dtype(dtype).kind == "f"
    x = np.zeros(maxdim, dtype=np.complex128)
    G = sd.Reflector(maxdim, np.dtype(dtype))
    lams, rs, ord_ = np.zeros(maxdim, dtype=np.complex128), np.zeros(maxdim), np.arange(maxdim)
    lt = sd.get_order(which)
    groups = np.zeros(maxdim, dtype=np.int64)
    stats, trail = {}, []
    active, k, prods = 0, mindim, mindim
    expand(A, st, 1, mindim, stats)
    for _ in range(restarts):
        expand(A, st, k + 1, maxdim, stats)
        prods += maxdim - k
        Q[:, :] = np.eye(maxdim, dtype=dtype)
        sd.local_schurfact(H[:maxdim, :], active, maxdim - 1, Q)
        ord_[:] = np.arange(maxdim)
        sd.copy_eigenvalues(lams, H)
        sd.copy_residuals(rs, H, Q, H[maxdim, maxdim - 1], x, active, maxdim - 1)
        sd.sort_perm(ord_, lams, lt)
        hfrob = float(np.linalg.norm(H))
        conv = lambda i: rs[i] <= max(sd.EPS * hfrob, tol * abs(lams[i]))  # noqa: E731
        eff = oa._include_conjugate_pair(real, lams, ord_, nev - 1) + 1
        nlock = 0
        for i in range(eff):
            if conv(ord_[i]):
                groups[ord_[i]] = 1
                nlock += 1
            else:
                groups[ord_[i]] = 2
        ideal = min(nlock + mindim, (mindim + maxdim) // 2)
        k, i = eff, eff
        while i < maxdim:
            pair = oa._include_conjugate_pair(real, lams, ord_, i) == i + 1
            if k < ideal and not conv(ord_[i]):
                grp = 2
                k += 2 if pair else 1
            else:
                grp = 3
            groups[ord_[i]] = grp
            if pair:
                groups[ord_[i + 1]] = grp
            i += 2 if pair else 1
        purge = 0
        while purge < active and groups[purge] == 1:
            purge += 1
        sd.partition_schur_three_way(H, Q, groups)
        sd.restore_arnoldi(H, nlock, k - 1, Q, G)
        # src/run.jl:363-365 in one T-folded product: S[:, 0:m+1) [ T[0:m, purge:m) Q[purge:m, purge:k) | T[:, m] ]
        m = maxdim
        Qe = np.zeros((m + 1, k - purge + 1), dtype=dtype)
        Qe[:m, : k - purge] = st.T[:m, purge:m] @ Q[purge:m, purge:k]
        Qe[:, k - purge] = st.T[: m + 1, m]
        st.S[:, purge : k + 1] = st.S[:, : m + 1] @ Qe
        st.T[:, :] = np.eye(m + 1, dtype=dtype)
        st.c = {}
        trail.append((k, nlock))
        active = nlock
        if active + 1 > nev:
            break
    nconv = active
    Q[:, :] = np.eye(maxdim, dtype=dtype)
    sd.sortschur(H, Q, nconv, lt)
    Vc = st.S[:, :nconv] @ Q[:nconv, :nconv]
    sd.copy_eigenvalues(lams, H, 0, nconv - 1)
    return Vc, H[:nconv, :nconv].copy(), lams[:nconv].copy(), prods, trail, stats


def _start(n, dtype, seed=3):
    v = oa.uniform_hash(seed, np.arange(n)).astype(dtype)
    if np.dtype(dtype).kind == "c":
        v = v + 1j * oa.uniform_hash(seed + 1, np.arange(n))
    return v


CASES = {
    "laplace-SR": (lambda: laplace3d(12, 13, 14), np.float64, dict(nev=6, which="SR", mindim=10, maxdim=24, tol=1e-10)),
    "laplace-cfg2-params": (lambda: laplace3d(14, 15, 16), np.float64, dict(nev=20, which="SR", mindim=20, maxdim=40, tol=1e-8)),
    "nonsymmetric-LM": (lambda: (sp.random(1500, 1500, density=5.0 / 1500, random_state=np.random.default_rng(3), format="csr")
                                 + sp.diags(np.linspace(1, 3, 1500))).tocsr(), np.float64,
                        dict(nev=8, which="LM", mindim=10, maxdim=20, tol=1e-9)),
    "complex-LM": (lambda: (laplace3d(9, 10, 11) + 1j * sp.diags(0.3 * np.cos(np.arange(990)))).tocsr().astype(np.complex128),
                   np.complex128, dict(nev=6, which="LM", mindim=10, maxdim=20, tol=1e-10)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_factored_expansion_reproduces_the_reference_order(case):
    build, dtype, kw = CASES[case]
    A = build()
    n = A.shape[0]
    v1 = _start(n, dtype)
    Vc, R, lam, prods, trail, stats = solve(A, v1, restarts=120, dtype=dtype, **kw)
    P, hist = oa.partialschur(A, v1=v1, restarts=120, **kw)
    assert prods == hist.mvproducts and len(lam) == hist.nconverged
    res, orth = np.linalg.norm(A @ Vc - Vc @ R), np.linalg.norm(Vc.conj().T @ Vc - np.eye(Vc.shape[1]))
    res0, orth0 = np.linalg.norm(A @ P.Q - P.Q @ P.R), np.linalg.norm(P.Q.conj().T @ P.Q - np.eye(P.Q.shape[1]))
    assert res <= 1.5 * res0 + 1e-13 and orth <= 3 * orth0 + 100 * EPS
    scale = np.abs(P.eigenvalues).max()
    assert np.abs(np.sort_complex(lam) - np.sort_complex(P.eigenvalues)).max() <= 1e-12 * scale


def test_the_delicate_case_low_rank_operator():
    """test/partial_schur.jl:6-27's matrix: after a near-breakdown the new basis vector lies in the null space of A, the
    product nearly cancels, and the first projection loses as many digits as y' - V g did.  With the DGKS test taken against
    max(||A v||, ||y'|| / beta) the implicit second pass restores orthogonality at EVERY step; with the reference's test
    alone it would end at 1e-3."""
    rng = np.random.default_rng(7)
    X = rng.random((10, 3))
    B = X @ X.T
    st = Factored(10, 7, np.float64)
    v1 = oa.uniform_hash(oa.DEFAULT_SEED, np.arange(10))
    st.S[:, 0] = v1 / np.linalg.norm(v1)
    stats = {}
    for j in range(1, 8):
        expand(B, st, j, j, stats)
        V = st.true_basis(j + 1)
        assert np.linalg.norm(V.T @ V - np.eye(j + 1)) < 50 * EPS * (j + 1), j
        assert np.linalg.norm(B @ V[:, :j] - V @ st.H[: j + 1, :j]) < 100 * EPS * np.linalg.norm(B), j
    Vc, R, lam, prods, trail, _ = solve(B, v1, 5, "LM", EPS, 5, 7, 50, np.float64)
    assert prods == 7 and len(lam) == 5                      # KAT-2: 7 products
    assert np.linalg.norm(Vc.T @ Vc - np.eye(5)) < 100 * EPS and np.linalg.norm(B @ Vc - Vc @ R) < 100 * EPS * np.linalg.norm(B)
