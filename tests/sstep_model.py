"""numpy MODEL of the s-step (block) Arnoldi expansion of the HIP path (csrc/ks_kernels.hpp: k_bdots / k_bupdate /
k_fin_blk; DESIGN.md section 3 "s-step expansion").  Test infrastructure: the executable statement of the algebra the
device kernels implement, checked against the oracle's reference-order expansion in tests/test_sstep_model.py.

The reference builds one Krylov vector per step (src/expansion.jl:116-133): operator, two passes over the basis for the
projection, two more when the DGKS test asks for the second projection.  The default device path needs two passes per
step (implicit second pass, tests/test_implicit_second_pass_model.py).  The s-step form needs two passes per BLOCK of s
steps:

    z_0 = S[:, k-1]                                   the last STORED column (coordinates u in the true basis: z_0 = V_k u)
    z_i = sigma_i (A z_{i-1} - theta_i z_{i-1})       i = 1..s: Newton basis, shifts theta = Leja-ordered Ritz values of the
                                                      previous restart, sigma_i powers of two (range only)
    P_raw = S_k^H Z                                   pass 1 over the basis, s right-hand sides          (k_bdots)
    P = T^H P_raw,  coef = T P                        true coordinates / coefficients of the stored columns
    W = Z - S_k coef                                  pass 2: first projection, written over Z           (k_bupdate)
    C_raw = S_k^H W,  G = W^H W                       ... which also delivers these (row-local once W is known)
    C = T^H C_raw                                     second-pass coefficients, NEVER applied to the n-vectors
    G'' = G - C^H C = R^H R                           Gram matrix of W'' = W - V_k C (V_k orthonormal), Cholesky
    T <- [[T, -T C R^-1], [0, R^-1]]                  V_true = S T stays the invariant; the new block is Q = W'' R^-1
    H columns k-1 .. k+s-2                            from the recurrence A z_{i-1} = z_i / sigma_i + theta_i z_{i-1}, the
                                                      coordinates zeta_i = [(P + C)[:, i]; R[:, i]] of z_i in V_{k+s}, and
                                                      the Arnoldi relation of the EARLIER columns (A V_{k-1} = V_k H)

Per block: s operator applications, the basis read TWICE (instead of 2 s times), two reductions (instead of s).
Breakdown / ill-conditioning: a Cholesky pivot below `pivot_min` times its diagonal entry abandons the block; the caller
redoes those steps one at a time (the per-step path takes the reference's breakdown decisions)."""
from __future__ import annotations

import numpy as np

from oracle import arnoldi as oa
from oracle import smalldense as sd

EPS = np.finfo(np.float64).eps


class BlockBail(Exception):
    def __init__(self, step, why):
        super().__init__(why)
        self.step = step


class Factored:
    """V_true = S T; u = coordinates of the last stored column in the true basis (length = number of columns)."""

    def __init__(self, n, maxdim, dtype):
        self.S = np.zeros((n, maxdim + 1), dtype=dtype, order="F")
        self.T = np.eye(maxdim + 1, dtype=dtype)
        self.H = np.zeros((maxdim + 1, maxdim), dtype=dtype, order="F")
        self.u = None          # None: the last column is an ordinary one (unit vector)
        self.diag = []         # per block: (k, s, cond(R), min pivot ratio)

    def true_basis(self, ncols):
        return self.S[:, :ncols] @ self.T[:ncols, :ncols]

    def materialize(self, ncols):
        self.S[:, :ncols] = self.true_basis(ncols)
        self.T[:, :] = np.eye(self.T.shape[0], dtype=self.T.dtype)
        self.u = None


def leja_order(pts):
    """Leja ordering of a point set (max |z| first, then maximise the product of distances to the chosen ones)."""
    pts = list(np.asarray(pts, dtype=np.complex128))
    if not pts:
        return []
    out = [pts.pop(int(np.argmax(np.abs(pts))))]
    logprod = np.zeros(len(pts))
    while pts:
        logprod += np.log(np.maximum(np.abs(np.asarray(pts) - out[-1]), 1e-300))
        i = int(np.argmax(logprod))
        out.append(pts.pop(i))
        logprod = np.delete(logprod, i)
    return out


def newton_shifts(ritz, s, real):
    """s shifts from the Ritz values of the previous restart.  Real arithmetic: real parts only (a conjugate pair
    contributes its real part once) -- the basis stays real, conditioning is what the bail test watches."""
    r = np.asarray(ritz, dtype=np.complex128)
    if real:
        r = np.unique(np.round(r.real, 14)).astype(np.complex128)
    lj = leja_order(r)
    if not lj:
        return np.zeros(s, dtype=np.float64 if real else np.complex128)
    th = [lj[i % len(lj)] for i in range(s)]
    return np.asarray(th).real.copy() if real else np.asarray(th)


def _pow2(x):
    return float(2.0 ** np.round(np.log2(x))) if np.isfinite(x) and x > 0 else 1.0


def dd_gram(W):
    """G = W^H W accumulated in double-double (what `gram="dd"` of the device kernel delivers): exact products via the
    fused multiply-add, compensated sums.  Modelled with longdouble / exact two-product sums; returns a float128-ish pair
    collapsed to (hi, lo)."""
    Wl = W.astype(np.clongdouble if np.iscomplexobj(W) else np.longdouble)
    G = Wl.conj().T @ Wl
    hi = G.astype(W.dtype)
    lo = (G - hi.astype(G.dtype)).astype(W.dtype)
    return hi, lo


def tri_inv(R):
    """Inverse of an upper triangular matrix by back substitution (works in extended precision too)."""
    s = R.shape[0]
    X = np.zeros_like(R)
    for c in range(s):
        X[c, c] = 1 / R[c, c]
        for i in range(c - 1, -1, -1):
            X[i, c] = -(R[i, i + 1:c + 1] @ X[i + 1:c + 1, c]) / R[i, i]
    return X


def chol_upper(G, pivot_min, k, j0):
    """R upper triangular with R^H R = G (Hermitian positive definite), bail when a pivot is below pivot_min * G[i,i]."""
    s = G.shape[0]
    R = np.zeros_like(G)
    ratios = []
    for i in range(s):
        d = G[i, i].real - np.vdot(R[:i, i], R[:i, i]).real
        ratio = d / G[i, i].real if G[i, i].real > 0 else 0.0
        ratios.append(ratio)
        if not (ratio > pivot_min):
            raise BlockBail(j0, f"pivot ratio {ratio:.2e} at block column {i} (k = {k})")
        R[i, i] = np.sqrt(d)
        for l in range(i + 1, s):
            R[i, l] = (G[i, l] - np.vdot(R[:i, i], R[:i, l])) / R[i, i]
    return R, min(ratios)


def expand_block(A, st, frm, to, shifts, s_max, stats, pivot_min=1e-6, gram="double", scale=None, combine=True):
    """Steps frm..to (step j builds 0-based column j) in blocks of at most s_max.  Raises BlockBail(step) with the state
    rolled back to the last complete block (columns < step are valid and T-lazy)."""
    S, T, H = st.S, st.T, st.H
    dtype = H.dtype
    j = frm
    blk = 0
    while j <= to:
        s = min(s_max, to - j + 1)
        k = j                                  # columns 0..k-1 exist, H[:k, :k-1] valid
        th = np.asarray([shifts[(blk * s_max + i) % len(shifts)] for i in range(s)], dtype=dtype)
        sig = np.full(s, scale if scale is not None else 1.0)
        u = st.u if st.u is not None else np.eye(k, dtype=dtype)[:, k - 1]
        assert len(u) == k
        # ---- s operator applications (Newton basis) ----
        Z = np.zeros((S.shape[0], s), dtype=dtype, order="F")
        prev = S[:, k - 1]
        for i in range(s):
            Z[:, i] = (A @ prev - th[i] * prev) * sig[i]
            prev = Z[:, i]
        # ---- pass 1 over the basis ----
        Praw = S[:, :k].conj().T @ Z
        P = T[:k, :k].conj().T @ Praw
        coef = T[:k, :k] @ P
        # ---- pass 2: first projection + what the second one needs ----
        W = Z - S[:, :k] @ coef
        Craw = S[:, :k].conj().T @ W
        if gram == "dd":
            Ghi, Glo = dd_gram(W)
        else:
            Ghi, Glo = W.conj().T @ W, None
        C = T[:k, :k].conj().T @ Craw
        if Glo is not None:
            Gpp = ((Ghi.astype(np.clongdouble) + Glo) - (C.conj().T @ C)).astype(np.clongdouble)
            try:
                Rl, minratio = chol_upper(Gpp if np.iscomplexobj(W) else Gpp.real.astype(np.longdouble), pivot_min, k, j)
            except BlockBail:
                raise
            R = Rl.astype(dtype)
        else:
            Gpp = Ghi - C.conj().T @ C
            R, minratio = chol_upper(Gpp, pivot_min, k, j)
        Rinv = tri_inv(R if Glo is None else Rl).astype(dtype)
        # ---- commit: stored columns, T ----
        S[:, k:k + s] = W
        T[:, k:k + s] = 0
        T[:k, k:k + s] = -(T[:k, :k] @ C) @ Rinv
        T[k:k + s, k:k + s] = Rinv
        # ---- H columns k-1 .. k+s-2 ----
        m = k + s
        zeta = np.zeros((m, s + 1), dtype=dtype)
        zeta[:k, 0] = u
        zeta[:k, 1:] = P + C
        zeta[k:, 1:] = R
        Hext = np.zeros((m, k), dtype=dtype)                      # A V_k = V_m Hext
        Hext[:k, :k - 1] = H[:k, :k - 1]
        a0 = zeta[:, 1] / sig[0] + th[0] * zeta[:, 0]             # A z_0
        a0[:k] -= H[:k, :k - 1] @ u[:k - 1]
        Hext[:, k - 1] = a0 / u[k - 1]
        if s > 1:
            rhs = zeta[:, 2:] / sig[1:][None, :] + zeta[:, 1:s] * th[1:][None, :]   # A z_i, i = 1..s-1
            rhs -= Hext @ (P + C)[:, :s - 1]
            M = np.linalg.solve(R[:s - 1, :s - 1].T, rhs.T).T      # rhs R_{s-1}^{-1}
        H[:m, k - 1] = Hext[:, k - 1]
        H[m:, k - 1] = 0
        for i in range(1, s):
            H[:m, k - 1 + i] = M[:, i - 1]
            H[m:, k - 1 + i] = 0
        if combine:
            # COMBINE: the last stored column becomes (up to the implicit second-pass part) the true last vector,
            #   S[:, m-1] <- W Rinv[:, s-1] = q_s + V_k (C Rinv[:, s-1])
            # one pass over the s columns of the block.  Without it the next block starts from W[:, s-1], whose components
            # along q_1..q_{s-1} feed into the next block's R: cond(R) then grows ~3x per block (6.7, 21, 75, 250, 640 on the
            # Laplacian) and the recovered H loses those digits.
            S[:, m - 1] = W @ Rinv[:, s - 1]
            T[k:m - 1, m - 1] = 0
            T[m - 1, m - 1] = 1
            cprime = C @ Rinv[:, s - 1]
            st.u = np.concatenate([cprime, np.zeros(s - 1, dtype=dtype), np.ones(1, dtype=dtype)])
        else:
            st.u = np.concatenate([C[:, s - 1], R[:, s - 1]])   # the last STORED column is W[:, s-1] = V_k C[:, s-1] + Q R[:, s-1]
        st.diag.append((k, s, float(np.linalg.cond(R / np.abs(np.diag(R))[None, :])), float(minratio)))
        stats["steps"] = stats.get("steps", 0) + s
        stats["blocks"] = stats.get("blocks", 0) + 1
        j += s
        blk += 1


def expand_block2(A, st, frm, to, shifts, s_max, stats, pivot_min=1e-6, scale=None, gdev_max=1e-8, inner=None, apply=None, ndefl=0, **_):
    """TWO-STAGE block step (what the device runs): the first pass also delivers G_Z = Z^H Z, so the block's triangular
    factor is known BEFORE the second pass, which then writes the block already (nearly) orthonormal:

        pass 1   P_raw = S_k^H Z,  G_Z = Z^H Z                          (k_bdots)
                 P = T^H P_raw;  G_1 = G_Z - P^H P = R_1^H R_1;  coef = T P        (||W||^2 = ||Z||^2 - ||P||^2, V_k orthonormal)
        pass 2   Qt = (Z - S_k coef) R_1^-1   written over Z;   C_raw = S_k^H Qt,  G_t = Qt^H Qt       (k_bupdate)
                 C = T^H C_raw;  G_2 = G_t - C^H C = R_2^H R_2  (~ I);  T <- [[T, -T C R_2^-1], [0, R_2^-1]]

    z_i = V_k (P + C R_1)[:, i] + Q (R_2 R_1)[:, i]: the H recovery is that of the one-stage form with R = R_2 R_1 and
    C_eff = C R_1.  The cancellation in G_1 only affects how close Qt is to orthonormal (G_t = I + delta); the second stage
    (a block classical Gram-Schmidt with Pythagorean inner products, applied twice) brings the orthogonality to rounding
    level as long as delta << 1.  The last stored column is q_s up to R_2 ~ I: no combine pass.

    IN-CHAIN DEFLATION (`ndefl` > 0; csrc/ks_block_kernels.hpp: k_defl_dots / k_defl_apply, HipBackend::defl_plan): every chain
    vector is projected against the locked columns U = V[:, 0:ndefl) as soon as it exists,
        z_i <- z_i - U c_i,  c_i = U^H z_i      so that      A z_{i-1} = z_i / sigma_i + theta_i z_{i-1} + U c_i / sigma_i
    and the H recovery adds c_i / sigma_i to the locked rows."""
    S, T, H = st.S, st.T, st.H
    dtype = H.dtype
    # `inner(X, Y)` = X^H Y and `apply(x)` = A x: the only places where the n-sized data is touched -- a row-partitioned run
    # passes an all-reduced inner product and a product with a halo exchange (tests/dist_worker.py), everything else is the
    # replicated small algebra
    inner = inner or (lambda X, Y: X.conj().T @ Y)
    apply = apply or (lambda x: A @ x)
    j = frm
    blk = 0
    while j <= to:
        s = min(s_max, to - j + 1)
        k = j
        th = np.asarray([shifts[(blk * s_max + i) % len(shifts)] for i in range(s)], dtype=dtype)
        sig = np.full(s, scale if scale is not None else 1.0)
        u = st.u if st.u is not None else np.eye(k, dtype=dtype)[:, k - 1]
        Z = np.zeros((S.shape[0], s), dtype=dtype, order="F")
        prev = S[:, k - 1]
        U = S[:, :ndefl] if ndefl > 0 else None     # (locked columns are ordinary ones: T = I there)
        cdefl = np.zeros((max(ndefl, 0), s), dtype=dtype)
        for i in range(s):
            Z[:, i] = (apply(prev) - th[i] * prev) * sig[i]
            if U is not None:
                cdefl[:, i] = inner(U, Z[:, i:i + 1])[:, 0]
                Z[:, i] -= U @ cdefl[:, i]
            prev = Z[:, i]
        # ---- pass 1 ----
        both = inner(np.hstack([S[:, :k], Z]), Z)      # ONE reduction: S^H Z and Z^H Z
        Praw, GZ = both[:k], both[k:]
        # a COLLAPSING chain: a step that shrinks the vector by a factor f puts an error of eps / f into the recovered H (the
        # recovery divides by the unscaled factor, whose pivots carry the chain's norms); the pivot test of chol_upper is relative
        # to each column's own norm and does not see it.  Below f = 3e-4 the block is abandoned (k_fin_blk stage 1)
        prev_n2 = 1.0
        for g_ in np.real(np.diag(GZ)):
            if not g_ > 1e-7 * prev_n2:
                raise BlockBail(j, f"Newton chain collapsed: norm^2 {g_:.1e} after {prev_n2:.1e} (k = {k})")
            prev_n2 = g_
        P = T[:k, :k].conj().T @ Praw
        R1, piv1 = chol_upper(GZ - P.conj().T @ P, pivot_min, k, j)
        R1inv = tri_inv(R1)
        coef = T[:k, :k] @ P
        # ---- pass 2 ----
        Qt = (Z - S[:, :k] @ coef) @ R1inv
        both = inner(np.hstack([S[:, :k], Qt]), Qt)    # ONE reduction: S^H Qt and Qt^H Qt
        Craw, Gt = both[:k], both[k:]
        C = T[:k, :k].conj().T @ Craw
        # G_t = I + delta with delta ~ eps cond(R_1)^2; the recovered H carries errors ~ eps cond(R_1): a block is accepted
        # only while delta <= gdev_max (1e-8: cond <= ~1e4, H as accurate as the per-step path's)
        gdev = float(np.abs(Gt - np.eye(s)).max())
        if not (gdev <= gdev_max):
            raise BlockBail(j, f"written block too far from orthonormal: {gdev:.1e} (k = {k})")
        R2, piv2 = chol_upper(Gt - C.conj().T @ C, 0.25, k, j)     # G_t ~ I: anything else means stage 1 failed
        R2inv = tri_inv(R2)
        S[:, k:k + s] = Qt
        T[:, k:k + s] = 0
        T[:k, k:k + s] = -(T[:k, :k] @ C) @ R2inv
        T[k:k + s, k:k + s] = R2inv
        # ---- H columns k-1 .. k+s-2 ----
        m = k + s
        R = R2 @ R1
        PC = P + C @ R1
        zeta = np.zeros((m, s + 1), dtype=dtype)
        zeta[:k, 0] = u
        zeta[:k, 1:] = PC
        zeta[k:, 1:] = R
        Hext = np.zeros((m, k), dtype=dtype)
        Hext[:k, :k - 1] = H[:k, :k - 1]
        a0 = zeta[:, 1] / sig[0] + th[0] * zeta[:, 0]
        a0[:ndefl] += cdefl[:, 0] / sig[0]
        a0[:k] -= H[:k, :k - 1] @ u[:k - 1]
        Hext[:, k - 1] = a0 / u[k - 1]
        H[:m, k - 1] = Hext[:, k - 1]
        H[m:, k - 1] = 0
        if s > 1:
            rhs = zeta[:, 2:] / sig[1:][None, :] + zeta[:, 1:s] * th[1:][None, :]
            rhs[:ndefl, :] += cdefl[:, 1:] / sig[1:][None, :]
            rhs -= Hext @ PC[:, :s - 1]
            M = np.linalg.solve(R[:s - 1, :s - 1].T, rhs.T).T
            for i in range(1, s):
                H[:m, k - 1 + i] = M[:, i - 1]
                H[m:, k - 1 + i] = 0
        st.u = np.concatenate([C[:, s - 1], R2[:, s - 1]])
        st.diag.append((k, s, float(np.linalg.cond(R1 / np.abs(np.diag(R1))[None, :])), float(min(piv1, piv2)),
                        float(np.linalg.norm(Gt - np.eye(s)))))
        stats["steps"] = stats.get("steps", 0) + s
        stats["blocks"] = stats.get("blocks", 0) + 1
        j += s
        blk += 1


def expand_steps(A, st, frm, to, stats):
    """Reference-order single steps on ORDINARY columns (what the per-step device path computes; used for the first
    expansion, which has no Ritz values to take shifts from, and after a bail)."""
    S, H = st.S, st.H
    n = S.shape[0]
    assert st.u is None
    for j in range(frm, to + 1):
        w = A @ S[:, j - 1]
        rnorm = np.linalg.norm(w)
        h = S[:, :j].conj().T @ w
        w = w - S[:, :j] @ h
        wnorm = np.linalg.norm(w)
        if wnorm < oa.ETA * rnorm:
            rnorm = wnorm
            c = S[:, :j].conj().T @ w
            w = w - S[:, :j] @ c
            h = h + c
            wnorm = np.linalg.norm(w)
            stats["reorth"] = stats.get("reorth", 0) + 1
        H[:j, j - 1] = h
        stats["steps"] = stats.get("steps", 0) + 1
        if wnorm <= oa.ETA * rnorm:
            H[j, j - 1] = 0.0
            if j != n:
                v = oa.uniform_hash(1000 + j, np.arange(n)).astype(H.dtype)
                for _ in range(2):
                    v -= S[:, :j] @ (S[:, :j].conj().T @ v)
                S[:, j] = v / np.linalg.norm(v)
                stats["breakdowns"] = stats.get("breakdowns", 0) + 1
            continue
        H[j, j - 1] = wnorm
        S[:, j] = w / wnorm


def defl_plan(H, j0, ritz, ratio=1.5, nmax=16, steps=10, tol=1.5e-8):
    """HipBackend::defl_plan (csrc/ks_backend.hpp): the leading locked columns (decoupled leading block of H: src/run.jl:330 zeroes
    the sub-diagonal entry behind it) whose eigenvalue exceeds `ratio` x the largest Ritz value of the rest.  Returns (number of
    columns, their eigenvalues)."""
    nl = 0
    for j in range(1, j0 - 1):
        if H[j, j - 1] == 0:
            nl = j
    if nl == 0 or ritz is None:
        return 0, []
    lam, width, j = [None] * nl, [1] * nl, 0
    while j < nl:
        if j + 1 < nl and H[j + 1, j] != 0:
            a, b, c, d = (complex(H[j, j]), complex(H[j, j + 1]), complex(H[j + 1, j]), complex(H[j + 1, j + 1]))
            tr2 = 0.5 * (a + d)
            disc = np.sqrt(tr2 * tr2 - (a * d - b * c) + 0j)
            lam[j], lam[j + 1] = tr2 + disc, tr2 - disc
            width[j], width[j + 1] = 2, 0
            j += 2
        else:
            lam[j] = complex(H[j, j])
            j += 1
    rest = 0.0
    for z in np.asarray(ritz, dtype=np.complex128):
        if not np.isfinite(abs(z)):
            continue
        if not any(abs(z - l) <= 1e-6 * max(1.0, abs(l)) for l in lam):
            rest = max(rest, abs(z))
    if not rest > 0:
        return 0, []
    nd, ex, j = 0, [], 0
    while j < nl:
        w = 2 if width[j] == 2 else 1
        mag = max(abs(lam[j]), abs(lam[j + 1])) if w == 2 else abs(lam[j])
        # (... and for which the growth matters over the block at hand: ratio^(steps - 1) above 1e3)
        if not mag > ratio * rest or not (mag / rest) ** max(1, steps - 1) > 1e3 or nd + w > nmax:
            break
        ex += lam[j:j + w]
        nd += w
        j += w
    return nd, ex


def expand(A, st, frm, to, stats, ritz, s, real, **kw):
    """What the backend does: blocks when shifts exist, single steps otherwise or after a bail."""
    nd, ex = 0, []
    lock_tol = kw.pop("lock_tol", 1.5e-8)
    if kw.pop("deflate", True) and ritz is not None:
        nd, ex = defl_plan(st.H, frm, ritz, steps=min(s, to - frm + 1), tol=lock_tol)
        if nd > stats.get("defl_last", 0):
            stats.pop("s_eff", None)           # (a problem that abandoned its blocks before those columns were locked gets its block size back)
        stats["defl_last"] = nd
    s = min(s, stats.get("s_eff", s))          # lowered after abandoned blocks (what the backend does: s -> s/2 -> 2 -> off)
    if s <= 1 or ritz is None:
        st.materialize(frm)
        expand_steps(A, st, frm, to, stats)
        return
    pool = np.asarray(ritz, dtype=np.complex128)
    if nd > 0:
        keep = np.asarray([not any(abs(z - q) <= 1e-6 * max(1.0, abs(q)) for q in ex) for z in pool])
        if keep.any():
            pool = pool[keep]
        kw["ndefl"] = nd
        stats["defl_blocks"] = stats.get("defl_blocks", 0) + 1
    shifts = newton_shifts(pool, s, real)
    scale = kw.pop("scale", None)
    if scale is None:
        rho = np.abs(pool).max()
        scale = 1.0 / _pow2(max(rho, 1e-300))
    variant = kw.pop("variant", "twostage")
    kw.pop("check", None)
    if variant != "twostage":
        kw.pop("ndefl", None)                  # (the one-stage form has no deflation)
    try:
        (expand_block2 if variant == "twostage" else expand_block)(A, st, frm, to, shifts, s, stats, scale=scale, **kw)
    except BlockBail as b:
        stats["bails"] = stats.get("bails", 0) + 1
        stats["s_eff"] = s // 2 if s >= 4 else (2 if s > 2 else 1)
        st.materialize(b.step)                 # the completed blocks stand (b.step = first step of the abandoned block)
        expand_steps(A, st, b.step, to, stats)


def solve(A, v1, nev, which, tol, mindim, maxdim, restarts, dtype, s=4, **kw):
    """oracle/arnoldi.py:_partialschur (src/run.jl:224-392) with the block expansion and the T-folded rotation."""
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
    ritz = None
    check = kw.pop("check", True)
    kw.setdefault("lock_tol", tol)             # (what the restart locks at: HipBackend::note_ritz hands it to defl_plan)
    expand(A, st, 1, mindim, stats, None, s, real)
    worst = dict(orth=0.0, rel=0.0)
    for _ in range(restarts):
        expand(A, st, k + 1, maxdim, stats, ritz, s, real, **kw)
        prods += maxdim - k
        if check:
            V = st.true_basis(maxdim + 1)
            worst["orth"] = max(worst["orth"], float(np.linalg.norm(V.conj().T @ V - np.eye(maxdim + 1))))
            worst["rel"] = max(worst["rel"], float(np.linalg.norm(A @ V[:, :maxdim] - V @ H) / np.linalg.norm(H)))
        Q[:, :] = np.eye(maxdim, dtype=dtype)
        sd.local_schurfact(H[:maxdim, :], active, maxdim - 1, Q)
        ord_[:] = np.arange(maxdim)
        sd.copy_eigenvalues(lams, H)
        ritz = lams.copy()
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
        # a selection that splits a 2 x 2 block of the real Schur form (complex pair not adjacent in the target's order) drops
        # the block's sub-diagonal entry here: the relation of the kept columns is violated by that much from now on (in the
        # reference too); blocks, which lean on it with O(1) coefficients, stay off for the rest of the run
        # (csrc/ks_driver.hpp RestartResult::leak, HipBackend::note_ritz)
        leak = float(np.abs(H[k:maxdim, :k]).max()) if k < maxdim else 0.0
        if leak > 1e-12 * hfrob:
            stats["relation_breaks"] = stats.get("relation_breaks", 0) + 1
            stats["s_eff"] = 1
        sd.restore_arnoldi(H, nlock, k - 1, Q, G)
        m = maxdim
        Qe = np.zeros((m + 1, k - purge + 1), dtype=dtype)
        Qe[:m, : k - purge] = st.T[:m, purge:m] @ Q[purge:m, purge:k]
        Qe[:, k - purge] = st.T[: m + 1, m]
        st.S[:, purge : k + 1] = st.S[:, : m + 1] @ Qe
        st.T[:, :] = np.eye(m + 1, dtype=dtype)
        st.u = None
        trail.append((k, nlock))
        active = nlock
        if active + 1 > nev:
            break
    nconv = active
    Q[:, :] = np.eye(maxdim, dtype=dtype)
    sd.sortschur(H, Q, nconv, lt)
    Vc = st.S[:, :nconv] @ Q[:nconv, :nconv]
    sd.copy_eigenvalues(lams, H, 0, nconv - 1)
    return dict(Q=Vc, R=H[:nconv, :nconv].copy(), eig=lams[:nconv].copy(), prods=prods, trail=trail, stats=stats, diag=st.diag, worst=worst)
