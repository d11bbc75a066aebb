"""`-m gpu`: seeded random sweep of the default expansion (implicit second DGKS pass, one reduction per step, T-folded restart
rotation) against the oracle: random sizes, Krylov dimensions up to 64, all five targets, both element types, symmetric /
nonsymmetric / clustered / rank-deficient operators (the last force breakdowns inside factored batches).  Acceptance per case:
where the trail is a well-posed quantity, same convergence flag and matrix-vector count as the oracle and Ritz values to 1e-8
relative; always, ||AQ - QR|| no worse than the oracle's own and ||Q'Q - I|| at rounding level (test/partial_schur.jl:104-105)."""
import numpy as np
import pytest
import scipy.sparse as sp

from __graft_entry__ import import_package
from oracle import arnoldi as oa

pytestmark = pytest.mark.gpu
pkg = import_package()
EPS = np.finfo(np.float64).eps


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(40, 1400))
    cplx = bool(rng.integers(0, 3) == 0)
    kind = ["sym", "nonsym", "cluster", "lowrank", "blockdiag"][int(rng.integers(0, 5))]
    dens = min(1.0, 6.0 / n)
    if kind == "sym":
        B = sp.random(n, n, density=dens, random_state=rng, format="csr")
        A = (B + B.T + sp.diags(rng.standard_normal(n) * 3)).tocsr()
    elif kind == "nonsym":
        A = (sp.random(n, n, density=dens, random_state=rng, format="csr") + sp.diags(np.linspace(1, 4, n))).tocsr()
    elif kind == "cluster":
        d = np.concatenate([np.full(n // 2, 1.0) + 1e-9 * rng.standard_normal(n // 2), np.linspace(2, 9, n - n // 2)])
        A = (sp.diags(d) + 1e-3 * sp.random(n, n, density=dens, random_state=rng, format="csr")).tocsr()
    elif kind == "lowrank":
        r = int(rng.integers(2, 6))
        X = rng.standard_normal((n, r))
        A = sp.csr_matrix(X @ X.T)
    else:  # an exactly invariant subspace reachable from the start vector: breakdown inside a batch
        m = int(rng.integers(4, 12))
        A = sp.block_diag([sp.csr_matrix(rng.standard_normal((m, m))), sp.diags(np.linspace(1, 2, n - m))], format="csr")
    if cplx:
        A = (A + 1j * sp.diags(0.2 * rng.standard_normal(n))).tocsr().astype(np.complex128)
    dtype = np.complex128 if cplx else np.float64
    maxdim = int(rng.integers(8, min(64, n - 1) + 1))
    nev = int(rng.integers(1, max(2, maxdim // 2)))
    mindim = int(rng.integers(nev, max(nev + 1, (nev + maxdim) // 2 + 1)))
    mindim = min(max(mindim, nev), maxdim)
    which = ["LM", "LR", "SR", "LI", "SI"][int(rng.integers(0, 5))]
    if not cplx and kind in ("sym", "cluster", "lowrank") and which in ("LI", "SI"):
        which = "SR"  # an imaginary-part target on a real spectrum orders +0.0 / -0.0: not a well-posed selection
    v1 = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    if kind == "blockdiag":
        v1 = np.zeros(n, dtype=dtype)
        v1[:m] = rng.standard_normal(m)
    return A.astype(dtype), v1.astype(dtype), dict(nev=nev, which=which, tol=1e-9, mindim=mindim, maxdim=maxdim, restarts=60), kind


_STRICT = {}   # seed -> did the case take the strict (trail-for-trail) branch; read by the population test at the end


# seeds 0..31, then ten more of the NON-NORMAL CLUSTER kind (hundreds of eigenvalues within 1e-9 of 1 under a 1e-3 non-symmetric
# perturbation: the family in which seed 17's drift of large blocks was found; seven Float64, three ComplexF64, Krylov dimensions
# 13-59, four different targets) -- each under the same acceptance: trail for trail where that is well posed, residual and
# orthogonality against the oracle's own always
CLUSTER_SEEDS = [33, 46, 53, 55, 58, 59, 64, 69, 85, 90]
# (in the suite: seven of them -- five Float64, two ComplexF64 --; the other three ran green all round and went when the suite's
# wall time on a slow box reached 928 s of the driver's 1 200: KS_FULL_SWEEP=1 brings them back)
import os as _os
_SUITE_CLUSTER_SEEDS = CLUSTER_SEEDS if _os.environ.get("KS_FULL_SWEEP") == "1" else CLUSTER_SEEDS[:7]
_OUTLIER_SEEDS = range(14) if _os.environ.get("KS_FULL_SWEEP") == "1" else range(10)


@pytest.mark.parametrize("seed", list(range(32)) + _SUITE_CLUSTER_SEEDS)
def test_random_case_against_the_oracle(seed):
    A, v1, kw, kind = _case(seed)
    ref, rh = oa.partialschur(A, v1=v1, **kw)
    dec, h = pkg.partialschur(A, v1=v1, **kw)
    tag = f"seed {seed} {kind} n={A.shape[0]} {A.dtype} {kw}: oracle {rh} device {h}"
    # where the comparison of TRAILS is meaningful: the Ritz values are not rounding noise (rank-deficient operators) and the
    # run is not a 60-restart non-convergent one (there the trail is a chaotic function of the last bits)
    well_posed = kind != "lowrank"
    # ... and the reference's OWN answer is a Schur pair to start with: degenerate parameter sets (mindim = nev = 1 across a
    # conjugate pair) make it "converge" to a pair with an O(1) residual -- a function of the last bits of every rounding, which
    # a differently-rounded but equally valid expansion (the s-step form, on by default) cannot and need not reproduce
    if rh.nconverged:
        res_ref0 = np.linalg.norm(A @ ref.Q - ref.Q @ ref.R)
        well_posed = well_posed and res_ref0 <= 1e-4 * max(1.0, sp.linalg.norm(A))
    settled = rh.converged and rh.restarts <= 40
    _STRICT[seed] = bool(well_posed and settled)
    if well_posed and settled:
        assert h.converged and h.nconverged == rh.nconverged and h.mvproducts == rh.mvproducts, tag
        scale = max(1.0, float(np.abs(ref.eigenvalues).max()))
        np.testing.assert_allclose(np.sort_complex(dec.eigenvalues), np.sort_complex(ref.eigenvalues), atol=1e-8 * scale, err_msg=tag)
    # always: whatever converged satisfies the reference's invariants as well as the oracle's own result does
    # (degenerate parameter sets -- mindim = nev = 1 across a conjugate pair -- make the REFERENCE return a poor pair: parity
    # includes that)
    if h.nconverged:
        Q, R = np.array(dec.Q), np.array(dec.R)
        nb = max(1.0, sp.linalg.norm(A))
        res = np.linalg.norm(A @ Q - Q @ R)
        res_ref = np.linalg.norm(A @ ref.Q - ref.Q @ ref.R) if rh.nconverged else 0.0
        assert res <= 10 * res_ref + 1e-8 * nb * max(1, h.nconverged), tag + f" residual {res:.2e} (oracle {res_ref:.2e})"
        assert np.linalg.norm(Q.conj().T @ Q - np.eye(Q.shape[1])) < 1e-11 * max(1, h.nconverged), tag


def _outlier_case(seed):
    """Block upper triangular [[D, C], [0, B]]: B a sparse random bulk of spectral radius ~1, D planted eigenvalues 6-60 x the bulk
    (1 x 1 and 2 x 2 blocks: exact eigenvalues of the whole), C an O(1) coupling -- the planted Schur vectors are NOT orthogonal to
    the rest of the spectrum's invariant subspace, which is what makes a Newton chain grow along them (in-chain deflation)."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(300, 2500))
    cplx = bool(rng.integers(0, 4) == 0)
    B = sp.random(n, n, density=6.0 / n, random_state=rng, format="csr", data_rvs=lambda k: rng.standard_normal(k) / np.sqrt(6.0))
    npl = int(rng.integers(1, 5))
    blocks, exact = [], []
    sign = -1.0 if rng.integers(0, 3) == 0 else 1.0          # (negative outliers: dominant in magnitude, the smallest real parts)
    for _ in range(npl):
        mag = float(rng.uniform(6.0, 60.0))
        if not cplx and rng.integers(0, 2) == 0:
            a, b = sign * mag * np.cos(0.4), mag * np.sin(0.4)
            blocks.append(np.array([[a, b], [-b, a]]))
            exact += [complex(a, b), complex(a, -b)]
        else:
            z = sign * mag * (np.exp(1j * rng.uniform(-0.5, 0.5)) if cplx else 1.0)
            blocks.append(np.array([[z]]))
            exact.append(complex(z))
    D = sp.block_diag(blocks, format="csr")
    k = D.shape[0]
    C = sp.csr_matrix(rng.standard_normal((k, n)) * (rng.random((k, n)) < 20.0 / n))
    A = sp.bmat([[D, C], [None, B]], format="csr")
    dtype = np.complex128 if cplx else np.float64
    if cplx:
        A = (A + 1j * sp.diags(np.concatenate([np.zeros(k), 0.1 * rng.standard_normal(n)]))).tocsr()
    A = A.astype(dtype)
    N = A.shape[0]
    maxdim = int(rng.integers(20, 41))
    nev = int(rng.integers(len(exact) + 1, len(exact) + 5))
    mindim = max(nev, maxdim // 2 - int(rng.integers(0, 4)))
    which = "LM" if sign > 0 and rng.integers(0, 2) == 0 else ("LR" if sign > 0 else "SR")
    v1 = rng.standard_normal(N) + (1j * rng.standard_normal(N) if cplx else 0)
    s_blk = [5, 10, 20][int(rng.integers(0, 3))]
    return A, v1.astype(dtype), dict(nev=nev, which=which, tol=1e-9, mindim=mindim, maxdim=maxdim, restarts=80), np.array(exact), s_blk


@pytest.mark.parametrize("seed", _OUTLIER_SEEDS)
def test_random_dominant_outliers_keep_their_blocks(seed):
    """Round 6c: random :LM / :LR / :SR problems whose wanted eigenvalues include planted outliers 6-60 x the bulk, coupled to it
    (non-normal), Float64 (incl. locked 2 x 2 blocks) and ComplexF64, blocks of 5 / 10 / 20: the planted values are found to 1e-8,
    the residual is no worse than the oracle's, the trail is the oracle's where it settles -- and the blocks stay ON: the chain is
    deflated against the locked outliers (ks_workspace_deflated_blocks) and at most one block (the first after a restart that locks
    a new outlier may still be probed too large) is abandoned, where rounds 3-5 ended such runs step by step."""
    A, v1, kw, exact, s_blk = _outlier_case(seed)
    ref, rh = oa.partialschur(A, v1=v1, **kw)
    ws = pkg.ArnoldiWorkspace(A.shape[0], kw["maxdim"], A.dtype)
    ws.set_sstep(s_blk)
    ws._v1 = v1
    dec, h = pkg.partialschur_(pkg.csr_operator(A), ws, **kw)
    info = ws.sstep_info
    tag = f"seed {seed} n={A.shape[0]} {A.dtype} s={s_blk} {kw} planted {np.round(exact, 2)}: oracle {rh} device {h} {info}"
    assert h.nconverged >= min(len(exact), rh.nconverged), tag
    lam = np.asarray(dec.eigenvalues)
    if h.nconverged >= len(exact):
        assert max(np.min(np.abs(lam - z)) for z in exact) <= 1e-8 * np.abs(exact).max(), tag
    if rh.converged and rh.restarts <= 50:
        assert h.converged and h.nconverged == rh.nconverged and h.mvproducts == rh.mvproducts, tag
    if h.nconverged:
        Q, R = np.array(dec.Q), np.array(dec.R)
        nb = max(1.0, sp.linalg.norm(A))
        res = np.linalg.norm(A @ Q - Q @ R)
        res_ref = np.linalg.norm(A @ ref.Q - ref.Q @ ref.R) if rh.nconverged else 0.0
        assert res <= 10 * res_ref + 1e-8 * nb * max(1, h.nconverged), tag + f" residual {res:.2e} (oracle {res_ref:.2e})"
        assert np.linalg.norm(Q.conj().T @ Q - np.eye(Q.shape[1])) < 1e-11 * max(1, h.nconverged), tag
    if info["blocks"] + info["abandoned"] > 0 and h.restarts >= 3:
        assert info["deflated_blocks"] > 0 and info["abandoned"] <= 1, tag
    ws.close()


def test_enough_cases_are_compared_trail_for_trail():
    """The sweep above compares mvproducts / nconverged / Ritz values with the oracle only where the trail is a well-posed
    quantity (see the comments there; the filter was widened in round 4 to admit the s-step default).  So that the filter
    cannot quietly swallow the sweep: 12 of the 32 seeds take the strict branch (measured with the oracle alone: seeds 0 1 4
    6 8 11 18 20 21 22 29 31); at least 10 must (two of slack for a BLAS that rounds the oracle differently)."""
    for sd in CLUSTER_SEEDS:
        assert _case(sd)[3] == "cluster", sd
    if len(_STRICT) < 32:
        pytest.skip("the sweep did not run in full in this process")
    strict = sorted(k for k, v in _STRICT.items() if v and k < 32)
    assert len(strict) >= 10, strict
