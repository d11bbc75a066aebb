"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement of the reference's small dense host kernels (layer L2 of
SURVEY.md): everything `_partialschur` does to the (maxdim+1) x maxdim
Hessenberg matrix `H` and the maxdim x maxdim accumulator `Q` between two
Arnoldi expansions.  Each function cites the reference file:line it follows.

All indices in THIS file are 0-based; a comment `# jl: ...` gives the 1-based
reference expression where the translation is not obvious.  Matrices are numpy
arrays of dtype float64 (real path) or complex128 (generic path).
"""
from __future__ import annotations

import cmath
import math

import numpy as np

from .givens import givens, givens_real

EPS = np.finfo(np.float64).eps


def is_real_dtype(a) -> bool:
    return np.asarray(a).dtype.kind == "f"


# --------------------------------------------------------------------------
# Rotations  (src/schurfact.jl:19-148)
# --------------------------------------------------------------------------
class Rotation2:
    """Givens rotation acting on rows/cols i, i+1.  src/schurfact.jl:19-23.

    Matrix form  [c s; -conj(s) c]  (src/schurfact.jl:38-46)."""

    __slots__ = ("c", "s", "i")

    def __init__(self, c, s, i):
        self.c, self.s, self.i = c, s, i

    def matrix(self, n, dtype):
        G = np.eye(n, dtype=dtype)
        i = self.i
        G[i, i] = self.c
        G[i + 1, i] = -np.conj(self.s)
        G[i, i + 1] = self.s
        G[i + 1, i + 1] = self.c
        return G


class Rotation3:
    """Two Givens rotations acting on rows i..i+2.  src/schurfact.jl:29-35, 48-52."""

    __slots__ = ("c1", "s1", "c2", "s2", "i")

    def __init__(self, c1, s1, c2, s2, i):
        self.c1, self.s1, self.c2, self.s2, self.i = c1, s1, c2, s2, i

    def matrix(self, n, dtype):
        G1 = Rotation2(self.c1, self.s1, self.i + 1).matrix(n, dtype)
        G2 = Rotation2(self.c2, self.s2, self.i).matrix(n, dtype)
        return G2 @ G1


def lmul(G, A, frm=None, to=None):
    """G * A on columns frm..to (inclusive).  src/schurfact.jl:76,82-101,121-134."""
    if A is None:  # NotWanted, src/schurfact.jl:74,79
        return
    if frm is None:
        frm, to = 0, A.shape[1] - 1
    if to < frm:
        return
    sl = slice(frm, to + 1)
    i = G.i
    if isinstance(G, Rotation3):
        a1 = A[i, sl].copy()
        a2 = A[i + 1, sl].copy()
        a3 = A[i + 2, sl].copy()
        a2p = G.c1 * a2 + G.s1 * a3
        a3p = -np.conj(G.s1) * a2 + G.c1 * a3
        a1pp = G.c2 * a1 + G.s2 * a2p
        a2pp = -np.conj(G.s2) * a1 + G.c2 * a2p
        A[i, sl] = a1pp
        A[i + 1, sl] = a2pp
        A[i + 2, sl] = a3p
    else:
        a1 = A[i, sl].copy()
        a2 = A[i + 1, sl].copy()
        A[i, sl] = G.c * a1 + G.s * a2
        A[i + 1, sl] = -np.conj(G.s) * a1 + G.c * a2


def rmul(A, G, frm=None, to=None):
    """A * G' on rows frm..to (inclusive).  src/schurfact.jl:77,103-119,136-148.

    NB: `rmul!` multiplies by the ADJOINT of the rotation (test/givens_rotation.jl:25-26)."""
    if A is None:
        return
    if frm is None:
        frm, to = 0, A.shape[0] - 1
    if to < frm:
        return
    sl = slice(frm, to + 1)
    i = G.i
    if isinstance(G, Rotation3):
        a1 = A[sl, i].copy()
        a2 = A[sl, i + 1].copy()
        a3 = A[sl, i + 2].copy()
        a2p = a2 * G.c1 + a3 * np.conj(G.s1)
        a3p = a2 * -G.s1 + a3 * G.c1
        a1pp = a1 * G.c2 + a2p * np.conj(G.s2)
        a2pp = a1 * -G.s2 + a2p * G.c2
        A[sl, i] = a1pp
        A[sl, i + 1] = a2pp
        A[sl, i + 2] = a3p
    else:
        a1 = A[sl, i].copy()
        a2 = A[sl, i + 1].copy()
        A[sl, i] = a1 * G.c + a2 * np.conj(G.s)
        A[sl, i + 1] = a1 * -G.s + a2 * G.c


def get_rotation2(p1, p2, i, real):
    """src/schurfact.jl:57-60."""
    c, s, nrm = givens(p1, p2, real)
    return Rotation2(c, s, i), nrm


def get_rotation3(p1, p2, p3, i, real):
    """src/schurfact.jl:65-69."""
    c1, s1, nrm1 = givens(p2, p3, real)
    c2, s2, nrm2 = givens(p1, nrm1, real)
    return Rotation3(c1, s1, c2, s2, i), nrm2


# --------------------------------------------------------------------------
# QR iterations on the active block (src/schurfact.jl)
# --------------------------------------------------------------------------
def is_offdiagonal_small(H, i, tol=EPS):
    """src/schurfact.jl:7-11 (i 0-based: tests H[i+1, i])."""
    return abs(H[i + 1, i]) <= tol * (abs(H[i, i]) + abs(H[i + 1, i + 1]))


def double_shift_schur(H, frm, to, trace, determinant, Q=None):
    """Francis double-shift bulge chase on H[frm..to, frm..to].  src/schurfact.jl:150-249."""
    m, n = H.shape
    H11 = H[frm, frm]
    H21 = H[frm + 1, frm]
    H12 = H[frm, frm + 1]
    H22 = H[frm + 1, frm + 1]
    H32 = H[frm + 2, frm + 1]

    p1 = H11 * H11 + H12 * H21 - trace * H11 + determinant
    p2 = H21 * (H11 + H22 - trace)
    p3 = H32 * H21

    G1, _ = get_rotation3(p1, p2, p3, frm, True)
    lmul(G1, H, frm, n - 1)
    rmul(H, G1, 0, min(frm + 3, m - 1))  # jl: rmul!(H, G1, 1, min(from+3, m))
    rmul(Q, G1)

    for i in range(frm + 1, to - 1):  # jl: from+1 : to-2
        p1 = H[i, i - 1]
        p2 = H[i + 1, i - 1]
        p3 = H[i + 2, i - 1]
        G, nrm = get_rotation3(p1, p2, p3, i, True)
        H[i, i - 1] = nrm
        H[i + 1, i - 1] = 0.0
        H[i + 2, i - 1] = 0.0
        lmul(G, H, i, n - 1)
        rmul(H, G, 0, min(i + 3, m - 1))
        rmul(Q, G)

    # last bulge: a single Givens rotation
    Gn, nrm = get_rotation2(H[to - 1, to - 2], H[to, to - 2], to - 1, True)
    H[to - 1, to - 2] = nrm
    H[to, to - 2] = 0.0
    lmul(Gn, H, to - 1, n - 1)
    rmul(H, Gn, 0, to)
    rmul(Q, Gn)
    return H


def single_shift_schur(H, frm, to, mu, Q=None):
    """Single-shift bulge chase.  src/schurfact.jl:251-320."""
    real = is_real_dtype(H)
    m, n = H.shape
    p1 = H[frm, frm] - mu
    p2 = H[frm + 1, frm]
    G1, _ = get_rotation2(p1, p2, frm, real)
    lmul(G1, H, frm, n - 1)
    rmul(H, G1, 0, min(frm + 2, m - 1))
    rmul(Q, G1)
    for i in range(frm + 1, to):  # jl: from+1 : to-1
        p1 = H[i, i - 1]
        p2 = H[i + 1, i - 1]
        G, nrm = get_rotation2(p1, p2, i, real)
        H[i, i - 1] = nrm
        H[i + 1, i - 1] = 0.0
        lmul(G, H, i, n - 1)
        rmul(H, G, 0, min(i + 2, m - 1))
        rmul(Q, G)
    return H


def _sign(x):
    return math.copysign(1.0, x) if x != 0.0 else 0.0


def upper_triangular_2x2(H11, H12, H21, H22):
    """(is_real, c, s): most stable rotation that triangularises a real 2x2 block.

    src/schurfact.jl:327-357; pinned by test/schurfact.jl:160-168."""
    if H21 == 0.0 or ((H11 - H22) == 0.0 and _sign(H12) != _sign(H21)):
        return False, 1.0, 0.0
    if H12 == 0.0:
        return True, 0.0, 1.0
    p = (H11 - H22) / 2
    bcmax = max(abs(H12), abs(H21))
    bcmis = min(abs(H12), abs(H21)) * _sign(H12) * _sign(H21)
    scale = max(abs(p), bcmax)
    z = (p / scale) * p + (bcmax / scale) * bcmis
    if z < 0:
        return False, 1.0, 0.0
    H11_min_lam = p + math.copysign(math.sqrt(scale) * math.sqrt(z), p)
    nrm = math.hypot(H21, H11_min_lam)
    return True, H11_min_lam / nrm, H21 / nrm


def use_single_shift(H11, H12, H21, H22):
    """(is_single, Wilkinson shift).  src/schurfact.jl:363-388; test/schurfact.jl:170-173."""
    scale = abs(H11) + abs(H12) + abs(H21) + abs(H22)
    H11 /= scale
    H12 /= scale
    H21 /= scale
    H22 /= scale
    t = (H11 + H22) / 2
    d = (H11 - t) * (H22 - t) - H12 * H21
    if d > 0.0:
        return False, 0.0
    sqrt_discr = math.sqrt(abs(d))
    l1 = t + sqrt_discr
    l2 = t - sqrt_discr
    lam = l1 if abs(H22 - l1) < abs(H22 - l2) else l2
    return True, lam * scale


class QRDidNotConverge(RuntimeError):
    """The reference throws the String "QR algorithm did not converge" (src/schurfact.jl:406)."""


def local_schurfact_real(H, start, to, Q=None, tol=EPS, maxiter=None):
    """Real quasi-triangularisation of H[start..to, start..to].  src/schurfact.jl:393-487."""
    if maxiter is None:
        maxiter = 100 * H.shape[0]
    it = 0
    ncols = H.shape[1]
    while to > start:
        it += 1
        if it > maxiter:
            raise QRDidNotConverge("QR algorithm did not converge")
        frm = to
        while frm > start:
            if is_offdiagonal_small(H, frm - 1, tol):
                H[frm, frm - 1] = 0.0
                break
            frm -= 1
        if frm == to:
            to -= 1
            continue
        C11, C12 = H[to - 1, to - 1], H[to - 1, to]
        C21, C22 = H[to, to - 1], H[to, to]
        if frm + 1 == to:
            is_real, cs, sn = upper_triangular_2x2(C11, C12, C21, C22)
            if is_real:
                G = Rotation2(cs, sn, frm)
                lmul(G, H, frm, ncols - 1)
                rmul(H, G, 0, to)
                rmul(Q, G)
                H[to, to - 1] = 0.0
            to -= 2
            continue
        is_single, mu = use_single_shift(C11, C12, C21, C22)
        if is_single:
            single_shift_schur(H, frm, to, mu, Q)
        else:
            trace = C11 + C22
            determinant = C11 * C22 - C12 * C21
            double_shift_schur(H, frm, to, trace, determinant, Q)
    return True


def local_schurfact_generic(H, start, to, Q=None, tol=EPS, maxiter=None):
    """Complex (generic) triangularisation, single shift only.  src/schurfact.jl:492-538.

    Returns False on non-convergence (the driver ignores it, src/run.jl:281)."""
    if maxiter is None:
        maxiter = 100 * H.shape[0]
    it = 0
    while True:
        it += 1
        if it > maxiter:
            return False
        frm = to
        while frm > start and not is_offdiagonal_small(H, frm - 1, tol):
            frm -= 1
        if frm == to:
            # jl: H[from, from-1] = zero(T).  The reference indexes out of bounds
            # (under @inbounds) when from == 1; guard that single case.
            if frm >= 1:
                H[frm, frm - 1] = 0.0
            to -= 1
        else:
            H11, H12 = H[to - 1, to - 1], H[to - 1, to]
            H21, H22 = H[to, to - 1], H[to, to]
            d = H11 * H22 - H21 * H12
            t = H11 + H22
            sqr = cmath.sqrt(t * t - 4 * d)
            l1 = (t + sqr) / 2
            l2 = (t - sqr) / 2
            lam = l1 if abs(H22 - l1) < abs(H22 - l2) else l2
            single_shift_schur(H, frm, to, lam, Q)
        if to <= start:
            break
    return True


def local_schurfact(H, start=None, to=None, Q=None, tol=EPS, maxiter=None):
    """Dispatch on arithmetic like the reference's two methods (+ the 4-arg convenience,
    src/schurfact.jl:540-545)."""
    if start is None:
        start, to = 0, H.shape[1] - 1
    if is_real_dtype(H):
        return local_schurfact_real(H, start, to, Q, tol, maxiter)
    return local_schurfact_generic(H, start, to, Q, tol, maxiter)


# --------------------------------------------------------------------------
# Eigenvalues of a quasi-triangular matrix (src/eigvals.jl:1-65)
# --------------------------------------------------------------------------
def copy_eigenvalues(lams, A, first=0, last=None, tol=EPS):
    """src/eigvals.jl:6-34.  Range first..last inclusive (0-based)."""
    if last is None:
        last = A.shape[1] - 1
    i = first
    while i < last:
        if is_offdiagonal_small(A, i, tol):
            lams[i] = A[i, i]
            i += 1
        else:
            d = A[i, i] * A[i + 1, i + 1] - A[i, i + 1] * A[i + 1, i]
            x = (A[i, i] + A[i + 1, i + 1]) / 2
            y = cmath.sqrt(complex(x * x - d))
            lams[i] = x + y
            lams[i + 1] = x - y
            i += 2
    if i == last:
        lams[i] = A[i, i]
    return lams


def eigenvalue(R, i):
    """src/eigvals.jl:42-55 (i points at the start of a block)."""
    n = min(R.shape)
    if i == n - 1 or R[i + 1, i] == 0:
        return complex(R[i, i])
    d = R[i, i] * R[i + 1, i + 1] - R[i, i + 1] * R[i + 1, i]
    x = (R[i, i] + R[i + 1, i + 1]) / 2
    y = cmath.sqrt(complex(x * x - d))
    return x + y


def eigenvalues(A, tol=EPS):
    """src/eigvals.jl:64-65."""
    return copy_eigenvalues(np.empty(A.shape[1], dtype=np.complex128), A, 0, A.shape[1] - 1, tol)


# --------------------------------------------------------------------------
# One eigenvector of a (quasi) upper triangular matrix
# (src/eigenvector_uppertriangular.jl)
# --------------------------------------------------------------------------
def shifted_backward_sub(x, R, lam, k, real):
    """Solve (R[0:k,0:k] - lam I) \\ x[0:k] in place; `k` = number of unknowns.

    src/eigenvector_uppertriangular.jl:6-42 (real quasi-triangular), :44-68 (generic)."""
    # jl k (1-based count) -> here kk = k-1 is the 0-based row being solved.
    while k > 0:
        kk = k - 1
        if real and k > 1 and R[kk, kk - 1] != 0:
            R11, R12 = R[kk - 1, kk - 1] - lam, R[kk - 1, kk]
            R21, R22 = R[kk, kk - 1], R[kk, kk] - lam
            det = R11 * R22 - R21 * R12
            a1 = (R22 * x[kk - 1] - R12 * x[kk]) / det
            a2 = (-R21 * x[kk - 1] + R11 * x[kk]) / det
            x[kk - 1] = a1
            x[kk] = a2
            for i in range(0, kk - 1):
                x[i] -= R[i, kk - 1] * x[kk - 1] + R[i, kk] * x[kk]
            k -= 2
        else:
            sigma = R[kk, kk] - lam
            if sigma == 0:
                x[kk] = sigma
            else:
                x[kk] /= sigma
                for i in range(0, kk):
                    x[i] -= R[i, kk] * x[kk]
            k -= 1
    return x


def collect_eigen(x, R, j):
    """Store the j-th (0-based) eigenvector of quasi-upper-triangular R in x[0:len];
    returns len.  src/eigenvector_uppertriangular.jl:76-129 (real), :131-154 (generic)."""
    real = is_real_dtype(R)
    n = R.shape[1]
    if real:
        if j < n - 1 and R[j + 1, j] != 0:
            j += 1
        if j > 0 and R[j, j - 1] != 0:
            R11, R21 = R[j - 1, j - 1], R[j, j - 1]
            R12, R22 = R[j - 1, j], R[j, j]
            det = R11 * R22 - R21 * R12
            tr = R11 + R22
            lam = (tr + cmath.sqrt(complex(tr * tr - 4 * det))) / 2
            x[j - 1] = -R12 / (R11 - lam)
            x[j] = 1.0
            for i in range(0, j - 1):
                x[i] = -R[i, j - 1] * x[j - 1] - R[i, j]
            shifted_backward_sub(x, R, lam, j - 1, True)  # jl: (x, R, λ, j-2) with 1-based j
        else:
            lam = R[j, j]
            x[j] = 1.0
            for i in range(0, j):
                x[i] = -R[i, j]
            shifted_backward_sub(x, R, lam, j, True)  # jl: (x, R, λ, j-1)
    else:
        lam = R[j, j]
        x[j] = 1.0
        for i in range(0, j):
            x[i] = -R[i, j]
        shifted_backward_sub(x, R, lam, j, False)
    nrm = 0.0
    for k in range(0, j + 1):
        nrm += abs(x[k]) ** 2
    scale = 1.0 / math.sqrt(nrm)
    for k in range(0, j + 1):
        x[k] *= scale
    return j + 1


def copy_residuals(rs, H, Q, h_last, x, first, last):
    """Ritz residual estimates |Q[m-1, :] . y| * |h_{m+1,m}|.  src/run.jl:524-545."""
    rs[:] = 0.0
    m = H.shape[1]
    for i in range(first, last + 1):
        x[:] = 0.0
        ln = collect_eigen(x, H, i)
        tmp = 0j
        for j in range(ln):
            tmp += Q[m - 1, j] * x[j]
        rs[i] = abs(tmp * h_last)
    return rs


# --------------------------------------------------------------------------
# Targets / ordering (src/targets.jl)
# --------------------------------------------------------------------------
def _isless(a: float, b: float) -> bool:
    """Julia `isless` on floats: total order, NaN last, -0.0 < 0.0."""
    if math.isnan(a):
        return False
    if math.isnan(b):
        return True
    if a == b:
        return math.copysign(1.0, a) < 0 and math.copysign(1.0, b) > 0
    return a < b


TARGETS = ("LM", "LR", "SR", "LI", "SI")


def get_order(which: str):
    """Return lt(a, b) on complex eigenvalues.  src/targets.jl:71-75."""
    if which == "LM":
        return lambda a, b: _isless(abs(b), abs(a))
    if which == "LR":
        return lambda a, b: _isless(b.real, a.real)
    if which == "SR":
        return lambda a, b: _isless(a.real, b.real)
    if which == "LI":
        return lambda a, b: _isless(b.imag, a.imag)
    if which == "SI":
        return lambda a, b: _isless(a.imag, b.imag)
    raise ValueError(f"Unknown target: {which}")  # ArgumentError, src/run.jl:185


def sort_perm(ord_, lams, lt):
    """Stable permutation sort, ties broken by index.  src/targets.jl:61-67, src/run.jl:289."""
    import functools

    def cmp(i, j):
        a, b = complex(lams[i]), complex(lams[j])
        if lt(a, b):
            return -1
        if lt(b, a):
            return 1
        return -1 if i < j else (1 if i > j else 0)

    ord_[:] = sorted(list(ord_), key=functools.cmp_to_key(cmp))
    return ord_


# --------------------------------------------------------------------------
# Reordering the Schur form (src/schursort.jl)
# --------------------------------------------------------------------------
def is_start_of_11_block(R, i):
    """src/schursort.jl:505."""
    return i == R.shape[1] - 1 or R[i + 1, i] == 0


def is_end_of_11_block(R, i):
    """src/schursort.jl:506."""
    return i == 0 or R[i, i - 1] == 0


def lu_complete_pivoting(A):
    """LU with complete pivoting of a tiny N x N system.  src/schursort.jl:79-140.

    Returns (LU, p, q, singular)."""
    A = np.array(A, copy=True)
    N = A.shape[0]
    p = [N - 1] * N
    q = [N - 1] * N
    singular = False
    for k in range(N - 1):
        m, n, maxval = 0, 0, 0.0
        # jl: for j = k:N, i = k:N  (j outer, i inner; strict '>' keeps the first max)
        for j in range(k, N):
            for i in range(k, N):
                if abs(A[i, j]) > maxval:
                    m, n, maxval = i, j, abs(A[i, j])
        p[k] = m
        q[k] = n
        for j in range(k, N):
            A[k, j], A[m, j] = A[m, j], A[k, j]
        for j in range(k, N):
            A[j, k], A[j, n] = A[j, n], A[j, k]
        Akk = A[k, k]
        if Akk == 0:
            singular = True
            break
        for i in range(k + 1, N):
            A[i, k] /= Akk
        for j in range(k + 1, N):
            Akj = A[k, j]
            for i in range(k + 1, N):
                A[i, j] -= A[i, k] * Akj
    if A[N - 1, N - 1] == 0:
        singular = True
    return A, p, q, singular


def lu_solve(LU, p, q, b):
    """src/schursort.jl:142-168."""
    x = np.array(b, copy=True)
    N = x.shape[0]
    for i in range(N):
        x[i], x[p[i]] = x[p[i]], x[i]
        for j in range(i + 1, N):
            x[j] -= LU[j, i] * x[i]
    for i in range(N - 1, -1, -1):
        for j in range(N - 1, i, -1):
            x[i] -= LU[i, j] * x[j]
        x[i] /= LU[i, i]
        x[i], x[q[i]] = x[q[i]], x[i]
    return x


def sylvsystem(A, B):
    """Kronecker form of A*X - X*B for 1x1 / 2x2 blocks.  src/schursort.jl:170-185."""
    dt = np.result_type(A, B)
    na, nb = A.shape[0], B.shape[0]
    if na == 1 and nb == 2:
        return np.array([[A[0, 0] - B[0, 0], -B[1, 0]], [-B[0, 1], A[0, 0] - B[1, 1]]], dtype=dt)
    if na == 2 and nb == 1:
        return np.array([[A[0, 0] - B[0, 0], A[0, 1]], [A[1, 0], A[1, 1] - B[0, 0]]], dtype=dt)
    if na == 2 and nb == 2:
        return np.array(
            [
                [A[0, 0] - B[0, 0], A[0, 1], -B[1, 0], 0],
                [A[1, 0], A[1, 1] - B[0, 0], 0, -B[1, 0]],
                [-B[0, 1], 0, A[0, 0] - B[1, 1], A[0, 1]],
                [0, -B[0, 1], A[1, 0], A[1, 1] - B[1, 1]],
            ],
            dtype=dt,
        )
    raise ValueError("sylvsystem: unsupported block sizes")


def sylv(A, B, C):
    """Solve A*X - X*B = C; returns (X, singular).  src/schursort.jl:198-202."""
    LU, p, q, singular = lu_complete_pivoting(sylvsystem(A, B))
    rhs = np.asarray(C).reshape(-1, order="F").astype(LU.dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = lu_solve(LU, p, q, rhs)
    return x.reshape(C.shape, order="F"), singular


def swap22_rotations(X, real):
    """src/schursort.jl:222-239."""
    c1, s1, nrm1 = givens(-X[1, 0], 1.0, real)
    c2, s2, nrm2 = givens(-X[0, 0], nrm1, real)
    X22 = c1 * -X[1, 1]
    X32 = -np.conj(s1) * -X[1, 1]
    X22 = -np.conj(s2) * -X[0, 1] + c2 * X22
    c3, s3, nrm3 = givens(X32, 1.0, real)
    c4, s4, nrm4 = givens(X22, nrm3, real)
    return c1, s1, c2, s2, c3, s3, c4, s4


def swap12_rotations(X, real):
    """src/schursort.jl:258-270."""
    c1, s1, _ = givens(-X[0, 0], 1.0, real)
    X22 = -np.conj(s1) * -X[0, 1]
    c2, s2, _ = givens(X22, 1.0, real)
    return c1, s1, c2, s2


def swap21_rotations(X, real):
    """src/schursort.jl:287-291."""
    c1, s1, nrm1 = givens(-X[1, 0], 1.0, real)
    c2, s2, _ = givens(-X[0, 0], nrm1, real)
    return c1, s1, c2, s2


def swap22(R, i, Q=None):
    """src/schursort.jl:307-350."""
    real = is_real_dtype(R)
    n = R.shape[1]
    A = R[i : i + 2, i : i + 2].copy()
    B = R[i + 2 : i + 4, i + 2 : i + 4].copy()
    C = R[i : i + 2, i + 2 : i + 4].copy()
    X, singular = sylv(A, B, C)
    if singular:
        return R
    c1, s1, c2, s2, c3, s3, c4, s4 = swap22_rotations(X, real)
    G1 = Rotation3(c1, s1, c2, s2, i)
    G2 = Rotation3(c3, s3, c4, s4, i + 1)
    lmul(G1, R, i, n - 1)
    rmul(R, G1, 0, i + 3)
    lmul(G2, R, i, n - 1)
    rmul(R, G2, 0, i + 3)
    R[i + 2, i] = 0
    R[i + 3, i] = 0
    R[i + 2, i + 1] = 0
    R[i + 3, i + 1] = 0
    rmul(Q, G1)
    rmul(Q, G2)
    return R


def swap21(R, i, Q=None):
    """2x2 block at i swapped with the 1x1 after it.  src/schursort.jl:365-401."""
    real = is_real_dtype(R)
    n = R.shape[1]
    A = R[i : i + 2, i : i + 2].copy()
    B = R[i + 2 : i + 3, i + 2 : i + 3].copy()
    C = R[i : i + 2, i + 2 : i + 3].copy()
    X, singular = sylv(A, B, C)
    if singular:
        return R
    c1, s1, c2, s2 = swap21_rotations(X, real)
    G1 = Rotation3(c1, s1, c2, s2, i)
    lmul(G1, R, i, n - 1)
    rmul(R, G1, 0, i + 2)
    R[i + 1, i] = 0
    R[i + 2, i] = 0
    rmul(Q, G1)
    return R


def swap12(R, i, Q=None):
    """1x1 block at i swapped with the 2x2 after it.  src/schursort.jl:419-458."""
    real = is_real_dtype(R)
    n = R.shape[1]
    A = R[i : i + 1, i : i + 1].copy()
    B = R[i + 1 : i + 3, i + 1 : i + 3].copy()
    C = R[i : i + 1, i + 1 : i + 3].copy()
    X, singular = sylv(A, B, C)
    if singular:
        return R
    c1, s1, c2, s2 = swap12_rotations(X, real)
    G1 = Rotation2(c1, s1, i)
    G2 = Rotation2(c2, s2, i + 1)
    lmul(G1, R, i, n - 1)
    rmul(R, G1, 0, i + 2)
    lmul(G2, R, i, n - 1)
    rmul(R, G2, 0, i + 2)
    R[i + 2, i] = 0
    R[i + 2, i + 1] = 0
    rmul(Q, G1)
    rmul(Q, G2)
    return R


def swap11(R, i, Q=None):
    """src/schursort.jl:460-482."""
    real = is_real_dtype(R)
    n = R.shape[1]
    R11 = R[i, i]
    R12 = R[i, i + 1]
    R22 = R[i + 1, i + 1]
    G, _ = get_rotation2(R12, R22 - R11, i, real)
    lmul(G, R, i + 2, n - 1)
    rmul(R, G, 0, i - 1)
    R[i, i] = R22
    R[i + 1, i + 1] = R11
    rmul(Q, G)
    return R


def swap(R, i, curr_11, next_11, Q=None):
    """src/schursort.jl:489-503."""
    if curr_11:
        if next_11:
            swap11(R, i, Q)
        else:
            swap12(R, i, Q)
    else:
        if next_11:
            swap21(R, i, Q)
        else:
            swap22(R, i, Q)


def rotate_right(R, frm, to, Q=None):
    """Move the block starting at `to` in front of the block at `frm`.  src/schursort.jl:19-32."""
    i = to
    while i > frm:
        curr_11 = is_start_of_11_block(R, i)
        prev_11 = is_end_of_11_block(R, i - 1)
        j = i - 1 if prev_11 else i - 2
        swap(R, j, prev_11, curr_11, Q)
        i = j


def partition_schur_three_way(R, Q, groups):
    """Stable three-way partition [1.. | 2.. | 3..].  src/run.jl:394-457."""
    hi = mi = lo = 0
    n = len(groups)
    while hi < n:
        group = groups[hi]
        blocksize = 1 if is_start_of_11_block(R, hi) else 2
        if group == 3:
            hi += blocksize
        elif group == 2:
            rotate_right(R, mi, hi, Q)
            hi += blocksize
            mi += blocksize
        else:
            rotate_right(R, lo, hi, Q)
            hi += blocksize
            mi += blocksize
            lo += blocksize


def sortschur(R, Q, to, lt):
    """Insertion sort of the first `to` (count) diagonal blocks.  src/run.jl:465-502."""
    if to <= 1:
        return
    next_idx = 0
    while next_idx <= to - 1:
        curr_idx = next_idx
        curr_size = 1 if is_start_of_11_block(R, curr_idx) else 2
        curr_lam = eigenvalue(R, curr_idx)
        while curr_idx > 0:
            prev_size = 1 if is_end_of_11_block(R, curr_idx - 1) else 2
            prev_idx = curr_idx - prev_size
            prev_lam = eigenvalue(R, prev_idx)
            if not lt(curr_lam, prev_lam):
                break
            swap(R, prev_idx, prev_size == 1, curr_size == 1, Q)
            curr_idx -= prev_size
        next_idx += curr_size


# --------------------------------------------------------------------------
# Restoring the Hessenberg form (src/restore_hessenberg.jl)
# --------------------------------------------------------------------------
def reflector(y, k):
    """Householder reflector from y[0:k] (k = length, pivot at y[k-1]); returns tau.

    src/restore_hessenberg.jl:16-45 (LAPACK clarfg-like).  The reference's guard at :18
    parses as `k <= 0 || (k > length(y) && return 0)`, i.e. it never returns early for
    k <= 0; callers only pass 2 <= k <= length(y)."""
    real = is_real_dtype(y)
    xnrm = 0.0
    for idx in range(k - 1):
        xnrm += abs(y[idx]) ** 2
    alpha = y[k - 1]
    if xnrm == 0.0 and (real or complex(alpha).imag == 0.0):
        return 0.0
    xnrm = math.sqrt(xnrm)
    if real:
        beta = -math.copysign(math.hypot(alpha, xnrm), alpha)
    else:
        beta = -math.copysign(math.hypot(abs(alpha), xnrm), complex(alpha).real)
    tau = (beta - alpha) / beta
    alpha = 1.0 / (alpha - beta)
    for i in range(k - 1):
        y[i] *= alpha
    y[k - 1] = beta
    return np.conj(tau)


class Reflector:
    """src/restore_hessenberg.jl:47-65."""

    def __init__(self, max_len, dtype):
        self.vec = np.zeros(max_len, dtype=dtype)
        self.offset = 0
        self.len = 0
        self.tau = 0.0

    def build(self, k):
        self.len = k
        self.tau = reflector(self.vec, k)
        return self.tau


def reflector_lmul(G, H, frm, to):
    """src/restore_hessenberg.jl:138-159."""
    ln, off, z, tau = G.len, G.offset, G.vec, G.tau
    if tau == 0:
        return
    for col in range(frm, to + 1):
        dot = 0.0
        for i in range(ln - 1):
            dot += np.conj(z[i]) * H[i + off, col]
        dot += H[ln - 1 + off, col]
        dot *= tau
        for i in range(ln - 1):
            H[i + off, col] -= dot * z[i]
        H[ln - 1 + off, col] -= dot


def reflector_rmul(H, G, frm, to):
    """src/restore_hessenberg.jl:161-182."""
    ln, off, z, tau = G.len, G.offset, G.vec, G.tau
    if tau == 0:
        return
    for row in range(frm, to + 1):
        dot = 0.0
        for i in range(ln - 1):
            dot += H[row, i + off] * z[i]
        dot += H[row, off + ln - 1]
        dot *= np.conj(tau)
        for i in range(ln - 1):
            H[row, i + off] -= dot * np.conj(z[i])
        H[row, off + ln - 1] -= dot


def restore_arnoldi(H, frm, to, Q, G):
    """Turn [R; h e_m' Q] back into Hessenberg form on columns frm..to (0-based, inclusive).

    src/restore_hessenberg.jl:75-134.  H is the full (m+1) x m array, Q is m x m."""
    if not frm < to:
        return
    real = is_real_dtype(H)
    m, n = H.shape  # m = maxdim + 1, n = maxdim
    nrm = Q[n - 1, frm]
    for i in range(frm, to):  # jl: from : to-1
        c, s, nrm = givens(Q[n - 1, i + 1], nrm, real)
        rot = Rotation2(c, -s, i)
        rmul(H, rot, 0, min(i + 2, to))
        lmul(rot, H, 0, to)
        rmul(Q, rot, 0, n - 1)
    H[to + 1, to] = Q[n - 1, to] * H[m - 1, n - 1]
    G.offset = frm
    for i in range(to - frm, 1, -1):  # jl: to-from : -1 : 2
        G.len = i
        row = frm + i
        for j in range(i):
            G.vec[j] = np.conj(H[row, j + frm])
        G.build(i)
        reflector_rmul(H, G, 0, row - 1)
        for j in range(i - 1):
            H[row, j + frm] = 0.0
        H[row, i - 1 + frm] = np.conj(G.vec[i - 1])
        reflector_lmul(G, H, frm, to)
        reflector_rmul(Q, G, 0, n - 1)
