"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement of the reference's Krylov-Schur driver and of the hot path it
drives: `ArnoldiWorkspace` (src/ArnoldiMethod.jl:41-93), `reinitialize!` /
`orthogonalize!` / `iterate_arnoldi!` (src/expansion.jl), `partialschur` /
`partialschur!` / `_partialschur` (src/run.jl:100-392) and `partialeigen`
(src/eigvals.jl:92-95).

The n-sized arithmetic is issued exactly as the reference issues it -- one
BLAS-2 / BLAS-1 call per verb -- through numpy/scipy, which dispatch into
OpenBLAS just as Julia's `LinearAlgebra` does (gemv 'T'/'C', gemv 'N', nrm2,
scal; gemm + two copies for the restart).  Summation order therefore differs
from any other implementation at rounding level; parity is defined by the
reference's own test invariants, not bitwise.

Parity pinned: by KAT-1..KAT-5 of SURVEY.md section 8c (tests/test_oracle_*.py).
Bitwise parity with Julia is unpinned (no Julia runtime in the build image and
the reference ships no golden vectors).

Indices are 0-based here; `# jl:` comments quote the 1-based reference line.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import smalldense as sd

ETA = math.sqrt(2.0) / 2.0  # "Constant used by ARPACK", src/expansion.jl:32,74


class DimensionMismatch(ValueError):
    """Julia's DimensionMismatch (thrown by `checksquare`, src/run.jl:110)."""


class ArgumentError(ValueError):
    """Julia's ArgumentError."""


# --------------------------------------------------------------------------
# portable RNG shared with the product
# --------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def splitmix64(x):
    """splitmix64 finaliser on uint64 numpy arrays."""
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
    return z ^ (z >> np.uint64(31))


def uniform_hash(seed: int, idx) -> np.ndarray:
    """u[i] = (splitmix64(seed xor i) >> 11) * 2^-53, uniform in [0,1)."""
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed & _M64) ^ np.asarray(idx, dtype=np.uint64))
    return (h >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def rand_fill(v, seed: int, row_offset: int = 0):
    """Counter-based stand-in for `rand!(v)` (src/expansion.jl:15): uniform [0,1) per real
    component, a pure function of (seed, global row index)."""
    n = v.shape[0]
    idx = np.arange(row_offset, row_offset + n, dtype=np.uint64)
    if v.dtype.kind == "c":
        v[:] = uniform_hash(seed, 2 * idx) + 1j * uniform_hash(seed, 2 * idx + 1)
    else:
        v[:] = uniform_hash(seed, idx)
    return v


def next_seed(seed: int, k: int) -> int:
    """Seed of the k-th random vector drawn in one solve."""
    return (seed + k * 0x9E3779B97F4A7C15) & _M64


DEFAULT_SEED = 20240917


# --------------------------------------------------------------------------
# operator seam
# --------------------------------------------------------------------------
def vtype(A):
    """src/run.jl:9-12: typeof(zero(T)/sqrt(one(T)))."""
    dt = np.dtype(getattr(A, "dtype", np.float64))
    return np.complex128 if dt.kind == "c" else np.float64


def checksquare(A):
    shp = A.shape
    if len(shp) != 2 or shp[0] != shp[1]:
        raise DimensionMismatch(f"matrix is not square: dimensions are {tuple(shp)}")
    return shp[0]


def apply_operator(A, y, x):
    """`mul!(y, A, x)` (src/expansion.jl:121)."""
    if hasattr(A, "mul_"):
        A.mul_(y, x)
    elif hasattr(A, "matvec") and not hasattr(A, "__matmul__"):
        y[:] = A.matvec(x)
    else:
        y[:] = A @ x


# --------------------------------------------------------------------------
# workspace (src/ArnoldiMethod.jl:41-93)
# --------------------------------------------------------------------------
class ArnoldiWorkspace:
    def __init__(self, V, H, V_tmp=None, Q=None):
        if V.shape[1] != H.shape[0]:
            raise ArgumentError("V should have the same number of columns as H has rows.")
        if H.shape[0] != H.shape[1] + 1:
            raise ArgumentError("H should have one more row than it has columns.")
        self.V = V
        self.H = H
        self.V_tmp = V_tmp if V_tmp is not None else np.empty_like(V)
        self.Q = Q if Q is not None else np.empty((H.shape[1], H.shape[1]), dtype=H.dtype, order="F")
        self.rng_seed = DEFAULT_SEED
        self.rng_count = 0

    @classmethod
    def from_dims(cls, dtype, matrix_order: int, krylov_dimension: int):
        """src/ArnoldiMethod.jl:60-69."""
        if not krylov_dimension <= matrix_order:
            raise ArgumentError("Krylov dimension should be less than matrix order.")
        V = np.zeros((matrix_order, krylov_dimension + 1), dtype=dtype, order="F")
        H = np.zeros((krylov_dimension + 1, krylov_dimension), dtype=dtype, order="F")
        return cls(V, H)

    @classmethod
    def from_vector(cls, v1, krylov_dimension: int):
        """src/ArnoldiMethod.jl:71-79."""
        dt = np.complex128 if np.asarray(v1).dtype.kind == "c" else np.float64
        V = np.zeros((len(v1), krylov_dimension + 1), dtype=dt, order="F")
        H = np.zeros((krylov_dimension + 1, krylov_dimension), dtype=dt, order="F")
        return cls(V, H)

    def default_populate(self, v):
        rand_fill(v, next_seed(self.rng_seed, self.rng_count))
        self.rng_count += 1


# --------------------------------------------------------------------------
# expansion (src/expansion.jl) -- THE HOT PATH
# --------------------------------------------------------------------------
def _nrm2(v):
    return float(np.linalg.norm(v))


def reinitialize(ws: ArnoldiWorkspace, j: int = 0, populate=None) -> bool:
    """Fill column j (0-based; jl column j+1) with a fresh vector orthonormal to V[:, :j].

    src/expansion.jl:12-59.  Does NOT touch H."""
    V = ws.V
    v = V[:, j]
    (populate or ws.default_populate)(v)
    rnorm = _nrm2(v)
    if j == 0:
        v /= rnorm  # jl: v ./= rnorm  (:28)
        return True
    Vprev = V[:, :j]
    h = Vprev.conj().T @ v  # :37
    v -= Vprev @ h  # :38
    wnorm = _nrm2(v)
    if wnorm < ETA * rnorm:  # :44
        rnorm = wnorm
        h = Vprev.conj().T @ v
        v -= Vprev @ h
        wnorm = _nrm2(v)
    if wnorm <= ETA * rnorm:  # :51
        return False
    v /= wnorm
    return True


def orthogonalize(ws: ArnoldiWorkspace, j: int, stats=None) -> bool:
    """DGKS classical Gram-Schmidt of column j (0-based) against columns 0..j-1;
    writes H[0:j, j-1] and H[j, j-1].   src/expansion.jl:69-109 (jl `j` = this j)."""
    V, H = ws.V, ws.H
    Vprev = V[:, :j]
    v = V[:, j]
    rnorm = _nrm2(v)  # :81
    h = Vprev.conj().T @ v  # :84  mul!(h, Vprev', v)
    v -= Vprev @ h  # :85  mul!(v, Vprev, h, -1, 1)
    wnorm = _nrm2(v)  # :88
    if wnorm < ETA * rnorm:  # :91
        rnorm = wnorm
        correction = Vprev.conj().T @ v
        v -= Vprev @ correction
        h = h + correction
        wnorm = _nrm2(v)
        if stats is not None:
            stats["reorth"] = stats.get("reorth", 0) + 1
    H[:j, j - 1] = h
    if wnorm <= ETA * rnorm:  # :99
        H[j, j - 1] = 0.0
        return False
    H[j, j - 1] = wnorm
    v /= wnorm  # :106
    return True


def iterate_arnoldi(A, ws: ArnoldiWorkspace, frm: int, to: int, stats=None):
    """jl: iterate_arnoldi!(A, arnoldi, from:to) with 1-based from/to (same numbers here:
    step j builds 0-based column j from column j-1).   src/expansion.jl:116-133."""
    V = ws.V
    n = V.shape[0]
    for j in range(frm, to + 1):
        apply_operator(A, V[:, j], V[:, j - 1])  # :121
        if stats is not None:
            stats["steps"] = stats.get("steps", 0) + 1
        if orthogonalize(ws, j, stats) is False and j != n:  # :127
            reinitialize(ws, j)
            if stats is not None:
                stats["breakdowns"] = stats.get("breakdowns", 0) + 1
    return ws


# --------------------------------------------------------------------------
# results
# --------------------------------------------------------------------------
@dataclass
class PartialSchur:
    """src/ArnoldiMethod.jl:130-137.  Q and R are VIEWS of the workspace (src/run.jl:149-150)."""

    Q: np.ndarray
    R: np.ndarray
    eigenvalues: np.ndarray


@dataclass
class History:
    """src/run.jl:217-222 (+ diagnostics the reference does not expose)."""

    mvproducts: int
    nconverged: int
    converged: bool
    nev: int
    restarts: int = 0

    def __str__(self):  # src/show.jl:3-21
        head = "Converged" if self.converged else "Not converged"
        return f"{head}: {self.nconverged} of {self.nev} eigenvalues in {self.mvproducts} matrix-vector products"


# --------------------------------------------------------------------------
# driver (src/run.jl)
# --------------------------------------------------------------------------
def _include_conjugate_pair(real, lams, ord_, i):
    """src/run.jl:510-517; i 0-based position in ord; returns i or i+1."""
    if not real:
        return i
    if i >= len(ord_) - 1:
        return i
    l1 = lams[ord_[i]]
    l2 = lams[ord_[i + 1]]
    return i + 1 if (l1.imag != 0 and np.conj(l1) == l2) else i


def _partialschur(A, ws, mindim, maxdim, nev, tol, restarts, which, active=0, stats=None, trace=None):
    """src/run.jl:224-392.  `active` 0-based (jl active-1)."""
    H, V, V_tmp, Q = ws.H, ws.V, ws.V_tmp, ws.Q
    dtype = H.dtype
    real = dtype.kind == "f"
    x = np.zeros(maxdim, dtype=np.complex128)
    G = sd.Reflector(maxdim, dtype)
    lams = np.zeros(maxdim, dtype=np.complex128)
    rs = np.zeros(maxdim, dtype=np.float64)
    ord_ = np.arange(maxdim)
    lt = sd.get_order(which)
    groups = np.zeros(maxdim, dtype=np.int64)

    k = mindim
    effective_nev = nev
    prods = len(range(active + 1, mindim + 1))  # jl: length(active:mindim), :264
    iterate_arnoldi(A, ws, active + 1, mindim, stats)  # :267
    nrestarts = 0

    def isconverged(i, Hfrob):  # :206-208
        return rs[i] <= max(sd.EPS * Hfrob, tol * abs(lams[i]))

    for _it in range(restarts):
        iterate_arnoldi(A, ws, k + 1, maxdim, stats)  # :272
        prods += len(range(k + 1, maxdim + 1))  # :275
        nrestarts += 1

        H_in = H.copy() if trace is not None else None
        Q[:, :] = np.eye(maxdim, dtype=dtype)  # :278
        Hm = H[:maxdim, :]
        sd.local_schurfact(Hm, active, maxdim - 1, Q)  # :281

        ord_[:] = np.arange(maxdim)
        sd.copy_eigenvalues(lams, H)  # :285
        sd.copy_residuals(rs, H, Q, H[maxdim, maxdim - 1], x, active, maxdim - 1)  # :286
        sd.sort_perm(ord_, lams, lt)  # :289
        Hfrob = float(np.linalg.norm(H))  # :292

        effective_nev = _include_conjugate_pair(real, lams, ord_, nev - 1) + 1  # :298
        nlock = 0
        for i in range(effective_nev):  # :301-308
            if isconverged(ord_[i], Hfrob):
                groups[ord_[i]] = 1
                nlock += 1
            else:
                groups[ord_[i]] = 2

        ideal_size = min(nlock + mindim, (mindim + maxdim) // 2)  # :316
        k = effective_nev
        i = effective_nev  # jl: i = effective_nev + 1 (1-based)
        while i < maxdim:  # :320
            is_pair = _include_conjugate_pair(real, lams, ord_, i) == i + 1
            num = 2 if is_pair else 1
            if k < ideal_size and not isconverged(ord_[i], Hfrob):
                group = 2
                k += num
            else:
                group = 3
            if is_pair:
                groups[ord_[i]] = group
                groups[ord_[i + 1]] = group
                i += 2
            else:
                groups[ord_[i]] = group
                i += 1

        purge = 0  # jl: purge = 1
        while purge < active and groups[purge] == 1:  # :351
            purge += 1

        if trace is not None:
            trace.append(
                dict(
                    iter=_it,
                    H_in=H_in,
                    active=active,
                    k=k,
                    nlock=nlock,
                    purge=purge,
                    effective_nev=effective_nev,
                    groups=groups.copy(),
                    lams=lams.copy(),
                    rs=rs.copy(),
                    ord=ord_.copy(),
                    H_schur=H.copy(),
                    Q_schur=Q.copy(),
                )
            )

        sd.partition_schur_three_way(H, Q, groups)  # :355
        sd.restore_arnoldi(H, nlock, k - 1, Q, G)  # :360  jl: (H, nlock+1, k, Q, G)

        # :363-365   (jl purge:k -> 0-based purge..k-1;  purge:maxdim -> purge..maxdim-1)
        V_tmp[:, purge:k] = V[:, purge:maxdim] @ Q[purge:maxdim, purge:k]
        V[:, purge:k] = V_tmp[:, purge:k]
        V[:, k] = V[:, maxdim]

        if trace is not None:
            trace[-1].update(H_after=H.copy(), Q_after=Q.copy())

        active = nlock  # jl: active = nlock + 1
        if active + 1 > nev:  # jl: active > nev
            break

    nconverged = active  # jl: active - 1
    Vconv = V[:, :nconverged]
    Hconv = H[:nconverged, :nconverged]

    Q[:, :] = np.eye(maxdim, dtype=dtype)
    sd.sortschur(H, Q, nconverged, lt)  # :379
    V_tmp[:, :nconverged] = Vconv @ Q[:nconverged, :nconverged]  # :382
    Vconv[:, :] = V_tmp[:, :nconverged]  # :383
    sd.copy_eigenvalues(lams, H, 0, nconverged - 1)  # :386

    history = History(prods, nconverged, nconverged >= nev, nev, nrestarts)
    schur = PartialSchur(Vconv, Hconv, lams[:nconverged].copy())
    return schur, history


def _check_which(which):
    if which not in sd.TARGETS:
        raise ArgumentError(f"Unknown target: {which}")
    return which


def partialschur(
    A,
    v1=None,
    nev=None,
    which="LM",
    tol=None,
    mindim=None,
    maxdim=None,
    restarts=200,
    seed=DEFAULT_SEED,
    stats=None,
    trace=None,
):
    """src/run.jl:100-129."""
    s = checksquare(A)
    n = s
    if nev is None:
        nev = min(6, n)
    if tol is None:
        tol = math.sqrt(sd.EPS)
    if mindim is None:
        mindim = min(max(10, nev), n)
    if maxdim is None:
        maxdim = min(max(20, 2 * nev), n)
    if nev < 1:
        raise ArgumentError("nev cannot be less than 1")
    if not (nev <= mindim <= maxdim <= s):
        raise ArgumentError(
            f"nev ≤ mindim ≤ maxdim ≤ size(A, 1) does not hold, got {nev} ≤ {mindim} ≤ {maxdim} ≤ {s}"
        )
    which = _check_which(which)
    if v1 is None:
        ws = ArnoldiWorkspace.from_dims(vtype(A), n, maxdim)
        ws.rng_seed = seed
        reinitialize(ws, 0)
    else:
        if len(v1) != n:
            raise ArgumentError("v1 should have the same dimension as A")
        ws = ArnoldiWorkspace.from_vector(v1, maxdim)
        ws.rng_seed = seed

        def _copy(v):
            v[:] = v1

        reinitialize(ws, 0, _copy)
    return _partialschur(A, ws, mindim, maxdim, nev, tol, restarts, which, 0, stats, trace)


def partialschur_(
    A,
    ws: ArnoldiWorkspace,
    start_from=1,
    initialize=None,
    nev=None,
    which="LM",
    tol=None,
    mindim=None,
    maxdim=None,
    restarts=200,
    stats=None,
    trace=None,
):
    """`partialschur!`, src/run.jl:152-179.  `start_from` is 1-based as in the reference."""
    s = checksquare(A)
    ncolsV = ws.V.shape[1]
    if initialize is None:
        initialize = start_from == 1
    if nev is None:
        nev = min(6, s)
    if tol is None:
        tol = math.sqrt(sd.EPS)
    if mindim is None:
        mindim = min(max(10, nev), s, ncolsV - 1)
    if maxdim is None:
        maxdim = min(max(20, 2 * nev), s, ncolsV - 1)
    if nev < 1:
        raise ArgumentError("nev cannot be less than 1")
    if not (nev <= mindim <= maxdim <= s):
        raise ArgumentError(
            f"nev ≤ mindim ≤ maxdim ≤ size(A, 1) does not hold, got {nev} ≤ {mindim} ≤ {maxdim} ≤ {s}"
        )
    if not maxdim < ncolsV:
        raise ArgumentError("maxdim should be strictly less than size(arnoldi.V, 2)")
    if not (1 <= start_from <= maxdim):
        raise ArgumentError("start_from should be between 1 and maxdim")
    which = _check_which(which)
    ws.H[:, start_from - 1 :] = 0  # :176
    if initialize:
        reinitialize(ws, start_from - 1)  # :177
    return _partialschur(A, ws, mindim, maxdim, nev, tol, restarts, which, start_from - 1, stats, trace)


def partialeigen(P: PartialSchur):
    """src/eigvals.jl:92-95: `eigen(P.R)` (LAPACK) then the tall-skinny `P.Q * vecs`."""
    import scipy.linalg as sla

    if P.R.shape[0] == 0:
        return np.zeros(0, dtype=np.complex128), np.zeros((P.Q.shape[0], 0), dtype=np.complex128)
    vals, vecs = sla.eig(P.R)
    return vals, P.Q @ vecs
