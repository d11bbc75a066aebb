"""`-m "not gpu"`: the numpy MODEL of the s-step (block) expansion (tests/sstep_model.py; device: csrc/ks_block.hip, DESIGN.md
section 3) against the oracle's reference-order expansion (oracle/arnoldi.py = src/expansion.jl:69-133, src/run.jl:224-392).

What is pinned here is the ALGORITHM the device kernels implement: Newton-basis blocks of s operator applications, the
basis read twice per block, two-stage block Gram-Schmidt with Pythagorean inner products, the Hessenberg columns recovered
from the basis recurrence.  Same Krylov space as the reference, so on well-posed problems: identical matrix-vector counts and
restart trails, Ritz values to 1e-10, residuals and orthogonality at the oracle's level -- for s = 2, 4, 5 (the sizes the
review asked for), 8, 10 and a few of the sizes in between and beyond (the device takes any size up to 20), BASELINE
configs 1-4 in miniature."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import arnoldi as oa
from oracle.matrices import laplace1d, laplace3d

import sstep_model as sm

EPS = np.finfo(np.float64).eps


def _start(n, dtype, seed=3):
    v = oa.uniform_hash(seed, np.arange(n)).astype(dtype)
    if np.dtype(dtype).kind == "c":
        v = v + 1j * oa.uniform_hash(seed + 1, np.arange(n))
    return v


CASES = {
    "config1-tridiagonal-SR": (lambda: laplace1d(100), np.float64, dict(nev=10, which="SR", mindim=10, maxdim=20, tol=1e-10)),
    "laplace-SR": (lambda: laplace3d(12, 13, 14), np.float64, dict(nev=6, which="SR", mindim=10, maxdim=24, tol=1e-10)),
    "config2-params": (lambda: laplace3d(14, 15, 16), np.float64, dict(nev=20, which="SR", mindim=20, maxdim=40, tol=1e-8)),
    "config3-nonsymmetric-LM": (lambda: (sp.random(1500, 1500, density=5.0 / 1500, random_state=np.random.default_rng(3), format="csr")
                                         + sp.diags(np.linspace(1, 3, 1500))).tocsr(), np.float64,
                                dict(nev=8, which="LM", mindim=10, maxdim=20, tol=1e-9)),
    "config4-complex-LM": (lambda: (laplace3d(9, 10, 11) + 1j * sp.diags(0.3 * np.cos(np.arange(990)))).tocsr().astype(np.complex128),
                           np.complex128, dict(nev=6, which="LM", mindim=10, maxdim=20, tol=1e-10)),
}


# (7, 9, 13, 19: the device's matrix-instruction kernels take the block size at run time -- round 5 --, so every size is one)
@pytest.mark.parametrize("s", [2, 4, 5, 7, 8, 9, 10, 13, 19])
@pytest.mark.parametrize("case", list(CASES))
def test_block_expansion_reproduces_the_reference(case, s):
    build, dtype, kw = CASES[case]
    A = build()
    v1 = _start(A.shape[0], dtype)
    r = sm.solve(A, v1, restarts=200, dtype=dtype, s=s, **kw)
    P, hist = oa.partialschur(A, v1=v1, restarts=200, **kw)
    assert r["stats"].get("blocks", 0) > 0 and r["stats"].get("bails", 0) == 0
    assert r["prods"] == hist.mvproducts and len(r["eig"]) == hist.nconverged       # same trail: same number of products
    Q, R = r["Q"], r["R"]
    res, orth = np.linalg.norm(A @ Q - Q @ R), np.linalg.norm(Q.conj().T @ Q - np.eye(Q.shape[1]))
    res0, orth0 = np.linalg.norm(A @ P.Q - P.Q @ P.R), np.linalg.norm(P.Q.conj().T @ P.Q - np.eye(P.Q.shape[1]))
    assert res <= 1.5 * res0 + 1e-12 and orth <= 3 * orth0 + 100 * EPS, (res, res0, orth, orth0)
    scale = np.abs(P.eigenvalues).max()
    assert np.abs(np.sort_complex(r["eig"]) - np.sort_complex(P.eigenvalues)).max() <= 1e-10 * scale
    # the invariants of test/expansion.jl:29-30 on EVERY cycle's full factorisation (true basis S T)
    assert r["worst"]["orth"] <= 1e-12 and r["worst"]["rel"] <= max(1e-12, 10 * kw["tol"])


def test_one_block_satisfies_the_arnoldi_relation_to_rounding():
    """A single batch, nothing locked: A V_m = V_{m+1} H and V'V = I to rounding; H equals the reference's H (implicit-Q:
    the Arnoldi factorisation of a start vector is unique up to signs, and both normalise with positive sub-diagonals)."""
    A = laplace3d(8, 9, 10)
    n = A.shape[0]
    v1 = _start(n, np.float64)
    ws = oa.ArnoldiWorkspace.from_dims(np.float64, n, 18)
    ws.V[:, 0] = v1 / np.linalg.norm(v1)
    oa.iterate_arnoldi(A, ws, 1, 18)
    for s in (2, 3, 5, 6):
        st = sm.Factored(n, 18, np.float64)
        st.S[:, 0] = v1 / np.linalg.norm(v1)
        stats = {}
        sm.expand_steps(A, st, 1, 6, stats)
        sm.expand_block2(A, st, 7, 18, sm.newton_shifts(np.linalg.eigvalsh(ws.H[:18, :18]), s, True), s, stats, scale=1 / 16)
        V = st.true_basis(19)
        assert np.linalg.norm(V.T @ V - np.eye(19)) <= 200 * EPS
        assert np.linalg.norm(A @ V[:, :18] - V @ st.H) <= 1e-13 * np.linalg.norm(st.H)
        assert np.abs(st.H - ws.H).max() <= 1e-12 * np.abs(ws.H).max(), s
        assert np.abs(V - ws.V).max() <= 1e-11


def test_breakdown_abandons_the_block_and_the_single_steps_take_over():
    """test/partial_schur.jl:6-27 (KAT-2): a rank-3 operator -- the Krylov space is exhausted after four vectors, the block's
    Gram matrix is singular, the Cholesky pivot test abandons the block and the per-step path (which takes the reference's
    breakdown decisions) redoes those steps: same 7 products, same five Ritz values, residual and orthogonality at rounding."""
    rng = np.random.default_rng(7)
    X = rng.random((10, 3))
    B = X @ X.T
    v1 = oa.uniform_hash(oa.DEFAULT_SEED, np.arange(10))
    r = sm.solve(B, v1, 5, "LM", EPS, 5, 7, 50, np.float64, s=2, check=False)
    P, hist = oa.partialschur(B, v1=v1, nev=5, which="LM", tol=EPS, mindim=5, maxdim=7, restarts=50)
    assert r["prods"] == hist.mvproducts == 7 and len(r["eig"]) == 5
    Q, R = r["Q"], r["R"]
    assert np.linalg.norm(Q.T @ Q - np.eye(5)) < 100 * EPS and np.linalg.norm(B @ Q - Q @ R) < 100 * EPS * np.linalg.norm(B)

    # an invariant subspace reached INSIDE a block of a longer run: block-diagonal operator, start vector in one block
    A = sp.block_diag([laplace1d(6), laplace1d(40) + 5 * sp.identity(40)]).tocsr()
    v = np.zeros(46)
    v[:6] = oa.uniform_hash(5, np.arange(6)) + 0.1
    st = sm.Factored(46, 12, np.float64)
    st.S[:, 0] = v / np.linalg.norm(v)
    stats = {}
    sm.expand_steps(A, st, 1, 3, stats)
    sm.expand(A, st, 4, 12, stats, np.linspace(0, 9, 12), 4, True)
    assert stats.get("bails", 0) == 1 and stats.get("breakdowns", 0) >= 1
    V = st.true_basis(13)
    assert np.linalg.norm(V.T @ V - np.eye(13)) < 1e-13
    assert np.linalg.norm(A @ V[:, :12] - V @ st.H) < 1e-13 * np.linalg.norm(st.H)


def test_leja_ordering_and_shift_selection():
    pts = [0.0, 1.0, 2.0, 3.0, 4.0]
    lj = sm.leja_order(pts)
    assert lj[0] == 4.0 and lj[1] == 0.0 and lj[2] == 2.0            # farthest point first, then alternately far from all chosen
    th = sm.newton_shifts(np.array([1 + 2j, 1 - 2j, 3.0, -1.0]), 3, True)
    assert th.dtype == np.float64 and set(th) == {1.0, 3.0, -1.0}    # a conjugate pair contributes its real part once
    thc = sm.newton_shifts(np.array([1 + 2j, 1 - 2j, 3.0]), 2, False)
    assert np.iscomplexobj(thc) and len(thc) == 2


def _split_pair_case(seed):
    """tests/test_gpu_factored_basis_stress.py::_ill_posed_case: a 1e-9-tight real cluster, a mildly non-symmetric
    perturbation and an imaginary-part target -- complex pairs whose members are not adjacent in the target's order."""
    rng = np.random.default_rng(4000 + seed)
    n = 300 + 37 * seed
    d = np.concatenate([np.full(n // 2, 1.0) + 1e-9 * rng.standard_normal(n // 2), np.linspace(2, 9, n - n // 2)])
    A = (sp.diags(d) + 1e-3 * sp.random(n, n, density=6.0 / n, random_state=rng, format="csr")).tocsr()
    v1 = rng.standard_normal(n)
    return A, v1, dict(nev=4, which=["LI", "SI"][seed % 2], tol=1e-9, mindim=8, maxdim=20, restarts=60)


@pytest.mark.parametrize("seed", [1, 3])
def test_blocks_stay_off_after_a_restart_that_split_a_conjugate_pair(seed):
    """Round 4 finding.  With an imaginary-part target on a real matrix the members of a complex pair are not neighbours in
    the sorted order, src/run.jl:298-339 does not see them as a pair, the truncation cuts through the 2 x 2 block of the real
    Schur form and drops its sub-diagonal entry: the Arnoldi relation of the kept columns is off by ~1e-3 ||A|| / ||H|| from
    the first restart on (5e-5 here) -- in the reference's own sequence (s = 1 below is the oracle's arithmetic).  The
    per-step expansion carries that error along unchanged.  The block expansion recovers H from the relation of the earlier
    columns with O(1) coefficients and feeds the result back restart after restart: without a guard the same runs reach
    3e-3 (s = 8) and 4e-2 (s = 10), on the device O(1).  Guard: the restart measures what it drops (zero for a proper
    Schur truncation) and blocks stay off from then on."""
    A, v1, kw = _split_pair_case(seed)
    base = sm.solve(A, v1, kw["nev"], kw["which"], kw["tol"], kw["mindim"], kw["maxdim"], kw["restarts"], np.float64, s=1)
    assert base["stats"].get("relation_breaks", 0) > 0          # the per-step run sees the same restarts
    for s in (5, 10):
        r = sm.solve(A, v1, kw["nev"], kw["which"], kw["tol"], kw["mindim"], kw["maxdim"], kw["restarts"], np.float64, s=s)
        assert r["stats"].get("relation_breaks", 0) > 0
        assert r["worst"]["rel"] <= 3.0 * base["worst"]["rel"], (s, r["worst"], base["worst"])
        assert r["worst"]["orth"] < 1e-12


def _disc_and_outlier(n=400, seed=5):
    """the operator of test/partial_schur.jl:122-138: a random matrix whose spectrum fills a disc of radius ~1, plus one eigenvalue
    at 50"""
    rng = np.random.default_rng(seed)
    D = rng.standard_normal((n, n)) / np.sqrt(n)
    D[0, 0] = 50.0
    return D


@pytest.mark.parametrize("case", ["disc-and-outlier", "planted-pairs"])
def test_in_chain_deflation_keeps_the_blocks_on_dominant_outliers(case):
    """Locked Schur vectors of DOMINANT eigenvalues (:LM with outliers) inside a block: the chain is orthogonalised against the
    basis only at the end of the block, by which time the components along those vectors (non-normal coupling, x |lambda_locked| /
    |lambda_rest| per step) -- or, with a shift at the locked value, the previous chain vector itself -- have made the projected
    columns parallel: the blocks of 10 are abandoned (3 times, down to single steps) on test/partial_schur.jl:122-138's operator.
    With the chain projected against those columns step by step and no shift at their eigenvalues (HipBackend::defl_plan,
    k_defl_dots / k_defl_apply, the extra term of k_fin_blk's H recovery): no block is abandoned, cond(R_1) <= 100 on that operator instead of
    3e3 (the abandoned ones: 1e14), the same products as the reference, and the relation of every cycle at rounding level."""
    if case == "disc-and-outlier":
        A, kw = _disc_and_outlier(), dict(nev=5, which="LM", mindim=10, maxdim=30, tol=1e-10)
    else:
        from oracle.matrices import hashed_nonsymmetric
        A = hashed_nonsymmetric(3000, seed=11, planted=[(30.0, 0.0), (25.0, 10.0), (-28.0, 0.0)])
        kw = dict(nev=6, which="LM", mindim=10, maxdim=30, tol=1e-10)
    v1 = oa.uniform_hash(20240917, np.arange(A.shape[0]))
    P, hist = oa.partialschur(A, v1=v1, restarts=100, **kw)
    off = sm.solve(A, v1, restarts=100, dtype=np.float64, s=10, deflate=False, **kw)
    on = sm.solve(A, v1, restarts=100, dtype=np.float64, s=10, **kw)
    assert off["stats"].get("bails", 0) >= 2                                          # what it cures
    assert on["stats"].get("bails", 0) == 0 and on["stats"].get("defl_blocks", 0) > 0, on["stats"]
    assert on["prods"] == hist.mvproducts and len(on["eig"]) == hist.nconverged
    assert max(d[2] for d in on["diag"]) <= (1e2 if case == "disc-and-outlier" else 1e4)   # cond(R_1) of every block
    assert on["worst"]["orth"] <= 1e-12 and on["worst"]["rel"] <= 1e-11, on["worst"]
    scale = np.abs(P.eigenvalues).max()
    assert np.abs(np.sort_complex(on["eig"]) - np.sort_complex(P.eigenvalues)).max() <= 1e-10 * scale
    Q, R = on["Q"], on["R"]
    assert np.linalg.norm(A @ Q - Q @ R) <= 10 * np.linalg.norm(A @ P.Q - P.Q @ P.R) + 1e-10


def test_deflation_plan_takes_the_leading_dominant_locked_columns():
    """sm.defl_plan = HipBackend::defl_plan (csrc/ks_backend.hpp): locked columns are the leading decoupled block of H; taken are
    the LEADING ones whose eigenvalue exceeds the largest other Ritz value by r > 1.5 with r^(steps - 1) > 1e3; a locked 2 x 2 block
    (conjugate pair of the real Schur form) is taken whole or not at all; locked values that do not dominate are left alone."""
    rng = np.random.default_rng(0)
    m = 12
    H = np.triu(rng.standard_normal((m + 1, m)), -1)
    H[:4, :4] = np.triu(H[:4, :4])
    H[0, 0] = 50.0
    H[1:3, 1:3] = [[30.0, 10.0], [-10.0, 30.0]]        # eigenvalues 30 +- 10 i
    H[3, 3] = 2.5
    H[4:, :4] = 0.0                                     # columns 0..3 are locked (decoupled): H[4, 3] == 0
    H[1, 0] = 0.0
    H[3, 2] = 0.0
    active = np.linalg.eigvals(H[4:m, 4:m])
    active = 1.2 * active / np.abs(active).max()        # the rest of the spectrum: radius 1.2
    ritz = np.concatenate([[50.0, 30 + 10j, 30 - 10j, 2.5], active])
    nd, ex = sm.defl_plan(H, m, ritz, steps=8)
    assert nd == 3 and len(ex) == 3                     # 50 and the pair; 2.5 is 2.1 x the rest: 2.1^7 = 170 < 1e3
    assert abs(ex[0] - 50.0) < 1e-12 and abs(abs(ex[1]) - np.hypot(30, 10)) < 1e-9 and abs(ex[1] - np.conj(ex[2])) < 1e-9
    nd20, _ = sm.defl_plan(H, m, ritz, steps=20)        # in a block of 20 the factor 2.1 matters too (2.1^19 = 1.3e6)
    assert nd20 == 4
    nd2, _ = sm.defl_plan(H, m, ritz, steps=2)          # a block of two steps: only the outlier at 50 (41.7 x the rest: 41.7 < 1e3)
    assert nd2 == 0
    # locked values that do NOT dominate (every :SR problem): nothing is deflated
    Hs = H.copy()
    Hs[0, 0], Hs[1:3, 1:3], Hs[3, 3] = 0.01, [[0.02, 0.0], [0.0, 0.03]], 0.04
    Hs[2, 1] = 0.0
    nd_sr, _ = sm.defl_plan(Hs, m, np.concatenate([[0.01, 0.02, 0.03, 0.04], active]), steps=20)
    assert nd_sr == 0
    # nothing locked: no plan
    Hn = H.copy()
    Hn[1, 0], Hn[3, 2], Hn[4, 3] = 0.3, 0.2, 0.1
    assert sm.defl_plan(Hn, m, ritz, steps=20)[0] == 0
