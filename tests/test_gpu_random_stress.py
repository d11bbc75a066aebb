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


@pytest.mark.parametrize("seed", list(range(32)) + CLUSTER_SEEDS)
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
