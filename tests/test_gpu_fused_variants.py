"""`-m gpu`: the fused, lazily normalised expansion in EVERY variant the library instantiates, against the oracle.

Round 1 pinned the fused kernel only up to 24 columns and only for Float64; the bench spends its time at
basis sizes 21..40 (k_axpy_dots_cs with 6..10 columns per wave) and config 4 is ComplexF64.  Here:

  * H and V of a full expansion to m = 24 / 40 / 64 columns, Float64 and ComplexF64, vs the oracle
    (src/expansion.jl:116-133) -- every column-per-wave count 1..16 of the fused kernel is crossed;
  * config 2's parameters (nev = 20, mindim = 20, maxdim = 40, :SR) end to end on the anisotropic
    30 x 31 x 32 Laplacian: identical mat-vec count and Ritz values to 1e-10 vs the oracle;
  * maxdim = 64 end to end (fused, 2 packs per lane) and maxdim = 65 (first eager size) for both types;
  * lazy ComplexF64 columns are materialised for every reader.
Everything goes through the C ABI (ctypes); tolerances next to each check.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from __graft_entry__ import import_package
from oracle import arnoldi as oa
from oracle.matrices import laplace3d, laplace3d_eigs

pytestmark = pytest.mark.gpu
pkg = import_package()
EPS = np.finfo(np.float64).eps
DTYPES = [np.float64, np.complex128]


def _operator(dtype, shape=(9, 10, 11)):
    """Laplacian (+ a skew-Hermitian-free complex perturbation for ComplexF64: keeps the spectrum off the real axis
    and makes the conjugation in V^H w observable)."""
    A = laplace3d(*shape)
    n = A.shape[0]
    if np.dtype(dtype).kind == "c":
        A = (A + 1j * sp.diags(0.25 * np.cos(np.arange(n))) + 0.1j * sp.diags(np.ones(n - 1), 1)).tocsr()
    return A.astype(dtype), n


def _start(dtype, n, seed=oa.DEFAULT_SEED):
    v = oa.uniform_hash(seed, np.arange(n))
    if np.dtype(dtype).kind == "c":
        v = v + 1j * oa.uniform_hash(seed + 1, np.arange(n))
    return v.astype(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m", [24, 40, 64])
def test_expansion_matches_oracle_H(dtype, m):
    """Same start vector, same operator: H built on the GPU equals the oracle's to 1e-11 and V to 1e-9 (continuous
    functions of the data while no DGKS branch flips -- the re-orthogonalisation counts must agree)."""
    A, n = _operator(dtype)
    v1 = _start(dtype, n)
    ows = oa.ArnoldiWorkspace.from_vector(v1, m)
    oa.reinitialize(ows, 0, lambda v: v.__setitem__(slice(None), v1))
    st = {}
    oa.iterate_arnoldi(A, ows, 1, m, st)
    op = pkg.csr_operator(A)
    ws = pkg.ArnoldiWorkspace(n, m, dtype)
    ws.reinitialize(0, v1)
    # two batches so that the second one starts on top of lazily normalised columns
    got1 = ws.iterate_arnoldi(op, 1, m // 2)
    got2 = ws.iterate_arnoldi(op, m // 2 + 1, m)
    assert got1["reorth"] + got2["reorth"] == st["reorth"] and got1["steps"] + got2["steps"] == m
    np.testing.assert_allclose(ws.H, ows.H, atol=1e-11)
    np.testing.assert_allclose(ws.V, ows.V, atol=1e-9)  # signs are fixed by H[j+1,j] > 0
    res, orth = ws.arnoldi_relation(op, m)
    assert res < 1e-12 * np.linalg.norm(ows.H) * 10 and orth < np.sqrt(EPS) / 100  # test/expansion.jl:29-30


def test_config2_parameters_end_to_end_vs_oracle():
    """BASELINE config 2's parameters (nev = 20, mindim = 20, maxdim = 40, :SR) at a size the oracle covers: every
    restart cycle runs basis sizes 21..40, i.e. exactly the fused-kernel instantiations the bench line is made of.
    Identical restart trail (mat-vec count) and Ritz values to 1e-10."""
    mx, my, mz = 30, 31, 32
    A = laplace3d(mx, my, mz)
    n = A.shape[0]
    v1 = _start(np.float64, n)
    kw = dict(nev=20, which="SR", tol=1e-10, mindim=20, maxdim=40, restarts=300)
    dec, hist = pkg.partialschur(A, v1=v1, **kw)
    ref, rhist = oa.partialschur(A, v1=v1, **kw)
    assert hist.converged and rhist.converged
    assert hist.mvproducts == rhist.mvproducts and hist.nconverged == rhist.nconverged
    np.testing.assert_allclose(np.sort(dec.eigenvalues.real), np.sort(ref.eigenvalues.real), atol=1e-10)
    np.testing.assert_allclose(np.sort(dec.eigenvalues.real)[:20], laplace3d_eigs(mx, my, mz)[:20], atol=1e-8)
    Q, R = dec.Q, np.array(dec.R)
    assert np.linalg.norm(A @ Q - Q @ R) < 1e-8 and np.linalg.norm(Q.T @ Q - np.eye(Q.shape[1])) < 100 * EPS * 20


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("maxdim", [64, 65])
def test_wide_fused_and_first_eager_size_end_to_end(dtype, maxdim):
    """maxdim = 64: the widest fused instantiation (16 columns per wave, inner products in two launches);
    maxdim = 65: the first size on the eager sequence.  Same decisions as the oracle either way."""
    rng = np.random.default_rng(33)
    n = 2100
    cplx = np.dtype(dtype).kind == "c"
    d = np.linspace(1, 60, n) + (0.2j * np.sin(np.arange(n)) if cplx else 0.0)
    P = sp.random(n, n, density=0.002, random_state=rng, format="csr", dtype=np.float64)
    A = (sp.diags(d) + 0.01 * (P + (1j * P.T if cplx else 0 * P))).tocsr().astype(dtype)
    v1 = _start(dtype, n, seed=5)
    kw = dict(nev=24, which="LR", tol=1e-9, mindim=32, maxdim=maxdim, restarts=100)
    dec, hist = pkg.partialschur(A, v1=v1, **kw)
    ref, rhist = oa.partialschur(A, v1=v1, **kw)
    assert hist.converged and hist.mvproducts == rhist.mvproducts and hist.nconverged == rhist.nconverged
    np.testing.assert_allclose(np.sort_complex(dec.eigenvalues), np.sort_complex(ref.eigenvalues), atol=1e-8)
    Q, R = dec.Q, np.array(dec.R)
    # per-vector criterion tol * |lambda| (src/run.jl:206-208), |lambda| <= 60, Frobenius norm over nconverged columns
    assert np.linalg.norm(A @ Q - Q @ R) < 10 * 1e-9 * 60 * np.sqrt(Q.shape[1])
    assert np.linalg.norm(Q.conj().T @ Q - np.eye(Q.shape[1])) < 100 * EPS * Q.shape[1]


def test_complex_shift_invert_parameters_vs_oracle():
    """Config 4's solver parameters (ComplexF64, nev = 6, mindim 10, maxdim 20, :LM) on a device-resident complex
    operator, so the whole expansion is the fused ComplexF64 sequence (the host-callback variant of config 4 is
    covered in test_gpu_parity.py)."""
    A, n = _operator(np.complex128, (12, 13, 14))
    v1 = _start(np.complex128, n, seed=11)
    kw = dict(nev=6, which="LM", tol=1e-10, mindim=10, maxdim=20, restarts=300)
    dec, hist = pkg.partialschur(A, v1=v1, **kw)
    ref, rhist = oa.partialschur(A, v1=v1, **kw)
    assert hist.converged and hist.mvproducts == rhist.mvproducts
    np.testing.assert_allclose(np.sort_complex(dec.eigenvalues), np.sort_complex(ref.eigenvalues), atol=1e-9)
    Q, R = dec.Q, np.array(dec.R)
    assert np.linalg.norm(A @ Q - Q @ R) < 1e-8
    dres, dorth = dec.workspace.residual_norms(pkg.as_operator(A), dec.nconverged)
    assert dres < 1e-8 and dorth < 100 * EPS * 6


def test_lazy_complex_columns_are_materialised_for_every_reader():
    A, n = _operator(np.complex128, (10, 11, 12))
    op = pkg.csr_operator(A)
    ws = pkg.ArnoldiWorkspace(n, 30, np.complex128)
    ws.reinitialize(0, _start(np.complex128, n, seed=7))
    ws.iterate_arnoldi(op, 1, 12)
    ws.iterate_arnoldi(op, 13, 30)
    H = np.array(ws.H)
    for j in (1, 5, 12, 13, 30):
        assert ws.norm(j) == pytest.approx(1.0, abs=1e-13)
    V = ws.V
    assert np.linalg.norm(V.conj().T @ V - np.eye(31)) < 1e-12
    np.testing.assert_allclose(A @ V[:, :30], V @ H, atol=1e-12)
    assert np.abs(ws.gemv_t(30, 30)).max() < 1e-12
    # a restart directly on top of lazy columns: the rotation absorbs the factors (rows of Q scaled)
    ws.reinitialize(0, _start(np.complex128, n, seed=8))
    ws.iterate_arnoldi(op, 1, 30)
    r = ws.restart(0, 6, "LM", 1e-10, 10, 30)
    Vr = ws.V
    k = r["k"]
    assert np.linalg.norm(Vr[:, : k + 1].conj().T @ Vr[:, : k + 1] - np.eye(k + 1)) < 1e-11
    Hr = np.array(ws.H)
    np.testing.assert_allclose(A @ Vr[:, :k], Vr[:, : k + 1] @ Hr[: k + 1, :k], atol=1e-10)


def test_operator_callback_exception_is_reraised():
    """A Python exception inside an operator callback cannot cross the C ABI; the wrapper must re-raise THAT
    exception (not a generic library error) from partialschur and from the verbs."""
    n = 200
    calls = {"n": 0}

    class Boom(RuntimeError):
        pass

    class Op:
        shape = (n, n)
        dtype = np.float64

        def mul_(self, y, x):
            calls["n"] += 1
            if calls["n"] == 7:
                raise Boom("seventh product fails")
            y[:] = np.arange(1, n + 1) * x

    with pytest.raises(Boom):
        pkg.partialschur(Op(), v1=np.ones(n), nev=3, which="LR", tol=1e-8)
    calls["n"] = 0
    op = pkg.as_operator(Op())
    ws = pkg.ArnoldiWorkspace(n, 20, ctx=op.ctx)
    ws.reinitialize(0, np.ones(n))
    with pytest.raises(Boom):
        ws.iterate_arnoldi(op, 1, 12)
    # the workspace is usable again after re-initialising (the aborted batch left no stale lazy bookkeeping)
    calls["n"] = 100
    ws.reinitialize(0, np.ones(n))
    st = ws.iterate_arnoldi(op, 1, 10)
    assert st["steps"] == 10
    V = ws.V
    assert np.linalg.norm(V[:, :11].T @ V[:, :11] - np.eye(11)) < 1e-12


@pytest.mark.parametrize("dtype", DTYPES)
def test_rotation_and_column_copy_verbs_on_lazy_columns(dtype):
    """ks_rotate / ks_col_copy called right after a fused expansion (what the Julia array-type seam does with the
    reference's own restart code, src/run.jl:363-365): the lazily normalised columns must be absorbed (rows of Q scaled,
    copy with the factor), not corrupted -- compare with numpy on the orthonormal basis."""
    A, n = _operator(dtype, (10, 11, 12))
    op = pkg.csr_operator(A)
    m = 24
    ws = pkg.ArnoldiWorkspace(n, m, dtype)
    ws.reinitialize(0, _start(dtype, n, seed=3))
    ws.iterate_arnoldi(op, 1, m)
    ws2 = pkg.ArnoldiWorkspace(n, m, dtype, ctx=ws.ctx)       # reference copy of the same basis, materialised
    ws2.reinitialize(0, _start(dtype, n, seed=3))
    ws2.iterate_arnoldi(op, 1, m)
    V = ws2.V                                                  # reading V materialises ws2's lazy columns
    rng = np.random.default_rng(12)
    c0, c, r = 2, m - 2, 9
    Qb = rng.standard_normal((c, r)) + (1j * rng.standard_normal((c, r)) if np.dtype(dtype).kind == "c" else 0)
    ws.rotate(c0, Qb.astype(dtype))                            # ws still has lazy columns 1..m
    ws.copy_col(c0 + r, m)
    got = ws.V
    want = V.copy()
    want[:, c0 : c0 + r] = V[:, c0 : c0 + c] @ Qb
    want[:, c0 + r] = V[:, m]
    np.testing.assert_allclose(got[:, : c0 + r + 1], want[:, : c0 + r + 1], atol=1e-12)
    # columns beyond the copied one are untouched orthonormal vectors (still valid after being materialised by the read)
    np.testing.assert_allclose(got[:, c0 + r + 1 :], V[:, c0 + r + 1 :], atol=1e-12)
    # and the expansion can continue on top: Arnoldi relation of the next steps holds w.r.t. the new basis
    ws.set_cols(0, np.linalg.qr(got[:, : c0 + r + 1])[0])
    ws.iterate_arnoldi(op, c0 + r + 1, m)
    Vn, H = ws.V, np.array(ws.H)
    assert np.linalg.norm(Vn.conj().T @ Vn - np.eye(m + 1)) < 1e-11
    for j in range(c0 + r + 1, m + 1):
        np.testing.assert_allclose(A @ Vn[:, j - 1], Vn[:, : j + 1] @ H[: j + 1, j - 1], atol=1e-11)


def test_config4_shift_invert_entirely_on_the_device():
    """BASELINE config 4 with the operator resident on the device: y = (A - sigma I)^{-1} x by rocSPARSE's pivoting
    tridiagonal solver plugged into the opaque DEVICE-operator seam (extras.TridiagonalShiftInvert) -- the vectors never
    cross PCIe.  Same interior eigenvalues as the host-LU callback run and as the dense spectrum; same mat-vec count as
    the oracle driven by the host LU (the two solvers agree to rounding, the restart trail must not notice)."""
    import scipy.sparse.linalg as spla

    from arnoldimethod_jl_amd import extras
    from oracle.matrices import laplace1d

    n = 400
    rng = np.random.default_rng(3)
    A = (laplace1d(n) + 1j * sp.diags(0.3 * rng.random(n))).tocsc().astype(np.complex128)
    sigma = 1.7 + 0.1j
    si = extras.TridiagonalShiftInvert(A.diagonal(-1), A.diagonal(0), A.diagonal(1), sigma)
    v1 = oa.uniform_hash(1, np.arange(n)) + 1j * oa.uniform_hash(2, np.arange(n))
    ws = pkg.ArnoldiWorkspace(v1, 20, ctx=si.operator.ctx)
    dec, hist = pkg.partialschur_(si.operator, ws, nev=6, which="LM", tol=1e-10, mindim=10, maxdim=20)
    lu = spla.splu((A - sigma * sp.identity(n)).tocsc())

    class HostLU:
        shape = (n, n)
        dtype = np.complex128

        def mul_(self, y, x):
            y[:] = lu.solve(x)

    ref, rhist = oa.partialschur(HostLU(), v1=v1, nev=6, which="LM", tol=1e-10, mindim=10, maxdim=20)
    assert hist.converged and hist.mvproducts == rhist.mvproducts
    lam = sigma + 1.0 / dec.eigenvalues
    exact = np.linalg.eigvals(A.toarray())
    want = exact[np.argsort(np.abs(exact - sigma))][:6]
    np.testing.assert_allclose(np.sort_complex(lam), np.sort_complex(want), atol=1e-8)
    np.testing.assert_allclose(np.sort_complex(dec.eigenvalues), np.sort_complex(ref.eigenvalues), atol=1e-8)
    # one application against the host LU
    x = rnd_c(rng, n)
    ws2 = pkg.ArnoldiWorkspace(n, 2, np.complex128, ctx=si.operator.ctx)
    ws2.set_col(0, x)
    ws2.apply(si.operator, 0, 1)
    np.testing.assert_allclose(ws2.col(1), lu.solve(x), rtol=1e-10, atol=1e-12)


def rnd_c(rng, n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)


@pytest.mark.parametrize("dtype", DTYPES)
def test_out_of_place_update_mode_is_bit_identical(dtype, monkeypatch):
    """KS_OOP=2 (default: product into a scratch vector, first projection out of place) and KS_OOP=1 (both projections out
    of place) are pure data movement: H and V of an expansion
    with and without second passes, and a whole solve, must equal the in-place run BIT for bit."""
    A, n = _operator(dtype, (11, 12, 13))
    v1 = _start(dtype, n, seed=21)
    out = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("KS_OOP", mode)
        op = pkg.csr_operator(A)
        ws = pkg.ArnoldiWorkspace(n, 30, dtype, ctx=op.ctx)
        ws.reinitialize(0, v1)
        st1 = ws.iterate_arnoldi(op, 1, 12)     # first batch: the 90 % rule is not armed yet (S0 only)
        st2 = ws.iterate_arnoldi(op, 13, 30)    # second batch: full out-of-place form
        H, V = np.array(ws.H), ws.V
        dec, hist = pkg.partialschur(A, v1=v1, nev=4, which="LM" if np.dtype(dtype).kind == "c" else "SR", tol=1e-10, maxdim=24)
        out[mode] = (st1, st2, H, V, hist.mvproducts, dec.eigenvalues.copy())
    for m2 in ("1", "2"):
        a, b = out["0"], out[m2]
        assert a[0] == b[0] and a[1] == b[1] and a[4] == b[4]
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[5], b[5])
    # a matrix that never takes the second pass (S1 is moved home by the update kernel when the rule was armed by a previous batch)
    rng = np.random.default_rng(2)
    B = (sp.random(n, n, density=5.0 / n, random_state=rng, format="csr") + 0 * sp.identity(n)).tocsr().astype(dtype)
    res = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("KS_OOP", mode)
        ws = pkg.ArnoldiWorkspace(n, 20, dtype)
        opA, opB = pkg.csr_operator(A, ws.ctx), pkg.csr_operator(B, ws.ctx)
        ws.reinitialize(0, v1)
        ws.iterate_arnoldi(opA, 1, 8)           # arms the rule (every step re-orthogonalises)
        s2 = ws.iterate_arnoldi(opB, 9, 20)     # random sparse operator: (almost) no second passes
        res[mode] = (s2, np.array(ws.H), ws.V)
    for m2 in ("1", "2"):
        assert res["0"][0] == res[m2][0] and np.array_equal(res["0"][1], res[m2][1]) and np.array_equal(res["0"][2], res[m2][2])
    assert res["1"][0]["reorth"] < 6


@pytest.mark.parametrize("passes", ["3", "2"])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["csr", "host_callback"])
def test_early_restart_handover_is_bit_identical(dtype, kind, passes, monkeypatch):
    """the part of the restart's host step that does not need H[maxdim+1, maxdim] (Schur form, Ritz values, unit
    residuals, ordering: src/run.jl:278-289) runs while the device finishes the last expansion step (KS_EARLY_RESTART=1,
    default).  It performs the same operations on the same numbers as the sequential order: every output of a whole solve
    -- eigenvalues, Q, R, restart count, products -- must be BIT-identical, with device operators (one batch per
    expansion) and with host callbacks (one batch per step)."""
    # (the early hand-over exists on the explicit-second-pass path only -- passes = 3 is where this test is not
    # vacuous; with the default two-pass expansion the switches must simply change nothing)
    monkeypatch.setenv("KS_PASSES", passes)
    A, n = _operator(dtype, (12, 13, 14))
    v1 = _start(dtype, n, seed=33)
    which = "LM" if np.dtype(dtype).kind == "c" else "SR"
    out = {}
    for mode in ("0", "1", "nomb"):
        monkeypatch.setenv("KS_EARLY_RESTART", "1" if mode == "nomb" else mode)
        monkeypatch.setenv("KS_MAILBOX", "0" if mode == "nomb" else "1")  # (read at workspace creation)
        if kind == "csr":
            op = A
        else:
            op = pkg.host_operator(lambda y, x: np.copyto(y, A @ x), n, dtype)
        dec, hist = pkg.partialschur(op, v1=v1, nev=6, which=which, tol=1e-10, mindim=10, maxdim=24, restarts=60)
        assert hist.converged and hist.restarts >= 3
        out[mode] = (dec.eigenvalues.copy(), np.array(dec.Q), np.array(dec.R), hist.mvproducts, hist.restarts, hist.nconverged)
    for m2 in ("1", "nomb"):
        a, b = out["0"], out[m2]
        assert a[3:] == b[3:]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("passes", ["3", "2"])
def test_early_restart_handover_with_breakdown_in_last_step(passes, monkeypatch):
    """An operator whose Krylov space is exhausted exactly at maxdim: the LAST step of the expansion breaks down (known only
    after the final reduction), the early part of the restart has already run on H and must be withdrawn (passes = 3: the
    path that hands H over early; passes = 2: nothing to withdraw, same results)."""
    monkeypatch.setenv("KS_PASSES", passes)
    n, m = 4000, 12
    # block-diagonal operator: an m x m block acting on the first m coordinates, identity elsewhere; start vector
    # supported on the first m coordinates -> invariant subspace of dimension m
    rng = np.random.default_rng(5)
    B = rng.standard_normal((m, m))
    A = sp.block_diag([sp.csr_matrix(B), sp.identity(n - m, format="csr")], format="csr")
    v1 = np.zeros(n)
    v1[:m] = rng.standard_normal(m)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("KS_EARLY_RESTART", mode)
        dec, hist = pkg.partialschur(A, v1=v1, nev=3, which="LM", tol=1e-10, mindim=6, maxdim=m, restarts=40)
        out[mode] = (dec.eigenvalues.copy(), hist.mvproducts, hist.restarts, hist.nconverged, hist.breakdowns)
    assert out["0"][1:] == out["1"][1:], (out["0"][1:], out["1"][1:])
    assert np.array_equal(out["0"][0], out["1"][0])
    assert out["1"][4] >= 1  # the breakdown did happen
    ref = np.linalg.eigvals(B)
    ref = ref[np.argsort(-np.abs(ref))][: len(out["1"][0])]
    assert np.allclose(np.sort_complex(out["1"][0]), np.sort_complex(ref), atol=1e-8)


@pytest.mark.parametrize("dtype", DTYPES)
def test_expand_restart_equals_the_two_separate_calls(dtype):
    """ks_expand_restart (one cycle of src/run.jl:272-365 in one call, early part of the host step overlapped with the tail of
    the expansion) leaves the workspace in the bit-identical state ks_iterate_arnoldi + ks_restart leave it in, cycle after
    cycle: H, V, basis size, locked count, Ritz values, residual estimates, groups."""
    A, n = _operator(dtype, (13, 12, 11))
    v1 = _start(dtype, n, seed=9)
    which = "LM" if np.dtype(dtype).kind == "c" else "SR"
    nev, mindim, maxdim = 5, 10, 22
    trails = []
    for fused in (False, True):
        op = pkg.csr_operator(A)
        ws = pkg.ArnoldiWorkspace(n, maxdim, dtype, ctx=op.ctx)
        ws.reinitialize(0, v1)
        ws.iterate_arnoldi(op, 1, mindim)
        k, active, trail = mindim, 0, []
        for _ in range(8):
            if fused:
                r = ws.expand_restart(op, k, active, nev, which, 1e-10, mindim, maxdim)
                assert r["steps"] == maxdim - k
            else:
                ws.iterate_arnoldi(op, k + 1, maxdim)
                r = ws.restart(active, nev, which, 1e-10, mindim, maxdim)
            k, active = r["k"], r["nlock"]
            trail.append((k, active, r["purge"], r["eigenvalues"].copy(), r["residuals"].copy(), r["groups"].copy(), np.array(ws.H), ws.cols(0, k + 1)))
            if active >= nev:
                break
        trails.append(trail)
    assert len(trails[0]) == len(trails[1]) >= 3
    for a, b in zip(*trails):
        assert a[:3] == b[:3]
        for x, y in zip(a[3:], b[3:]):
            assert np.array_equal(x, y)


# ------------------------------------------------------------------------------------------------------------------
# implicit second pass (KS_PASSES=2, default) vs the second projection applied to the vector (KS_PASSES=3)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_two_and_three_basis_passes_agree(dtype, monkeypatch):
    """The default expansion carries the DGKS second projection (src/expansion.jl:93-94) in a triangular factor T
    (V = S T, two reads of the basis per step); KS_PASSES=3 applies it to the vector as the reference does.  Same
    decisions, same H to 1e-12 (relative to ||H||), same orthonormal basis to 1e-10 once T is folded in, and whole solves
    with identical restart trails."""
    A, n = _operator(dtype, (11, 12, 13))
    v1 = _start(dtype, n, seed=17)
    which = "LM" if np.dtype(dtype).kind == "c" else "SR"
    out = {}
    for passes in ("3", "2"):
        monkeypatch.setenv("KS_PASSES", passes)
        op = pkg.csr_operator(A)
        ws = pkg.ArnoldiWorkspace(n, 40, dtype, ctx=op.ctx)
        assert ws.passes == int(passes)
        ws.reinitialize(0, v1)
        s1 = ws.iterate_arnoldi(op, 1, 13)
        s2 = ws.iterate_arnoldi(op, 14, 40)       # continues on top of columns that are still in factored form
        H = np.array(ws.H)
        res, orth = ws.arnoldi_relation(op, 40)   # materialises: V_true = S T
        assert res < 1e-11 * np.linalg.norm(H) and orth < np.sqrt(EPS) / 100
        dec, hist = pkg.partialschur(A, v1=v1, nev=6, which=which, tol=1e-10, mindim=12, maxdim=26, restarts=200)
        assert hist.converged
        out[passes] = (s1, s2, H, ws.V, hist.mvproducts, hist.restarts, np.sort_complex(dec.eigenvalues))
    a, b = out["3"], out["2"]
    assert a[0] == b[0] and a[1] == b[1] and a[4:6] == b[4:6]
    hn = np.linalg.norm(a[2])
    assert np.linalg.norm(a[2] - b[2]) < 1e-12 * hn
    np.testing.assert_allclose(a[3], b[3], atol=1e-10)
    np.testing.assert_allclose(a[6], b[6], atol=1e-11 * hn)


@pytest.mark.parametrize("dtype", DTYPES)
def test_three_pass_path_still_matches_the_oracle(dtype, monkeypatch):
    """KS_PASSES=3 (the round-2 expansion: second projection applied to the vector, lazy normalisation by column factors)
    stays covered: H / V against the oracle, the lazy-aware verbs, a whole solve."""
    monkeypatch.setenv("KS_PASSES", "3")
    A, n = _operator(dtype)
    v1 = _start(dtype, n)
    m = 24
    ows = oa.ArnoldiWorkspace.from_vector(v1, m)
    oa.reinitialize(ows, 0, lambda v: v.__setitem__(slice(None), v1))
    st = {}
    oa.iterate_arnoldi(A, ows, 1, m, st)
    op = pkg.csr_operator(A)
    ws = pkg.ArnoldiWorkspace(n, m, dtype, ctx=op.ctx)
    assert ws.passes == 3
    ws.reinitialize(0, v1)
    g1 = ws.iterate_arnoldi(op, 1, 9)
    g2 = ws.iterate_arnoldi(op, 10, m)
    assert g1["reorth"] + g2["reorth"] == st["reorth"]
    np.testing.assert_allclose(ws.H, ows.H, atol=1e-11)
    np.testing.assert_allclose(ws.V, ows.V, atol=1e-9)
    which = "LM" if np.dtype(dtype).kind == "c" else "SR"
    dec, hist = pkg.partialschur(A, v1=v1, nev=5, which=which, tol=1e-10, mindim=10, maxdim=22)
    ref, rhist = oa.partialschur(A, v1=v1, nev=5, which=which, tol=1e-10, mindim=10, maxdim=22)
    assert hist.converged and hist.mvproducts == rhist.mvproducts
    np.testing.assert_allclose(np.sort_complex(dec.eigenvalues), np.sort_complex(ref.eigenvalues), atol=1e-10)


def test_implicit_second_pass_when_the_correction_is_not_small():
    """Where the implicit form is delicate: a near-breakdown step leaves a column whose second-pass correction is not small
    against it, and the next product nearly cancels (the new basis vector lies in the null space of a low-rank operator,
    test/partial_schur.jl:6-27).  The DGKS test is then taken against the norm of the vector the projection really worked
    on; orthogonality and the Arnoldi relation must hold to working precision at EVERY step."""
    rng = np.random.default_rng(7)
    for n, rank in ((10, 3), (60, 5), (400, 7)):
        X = rng.random((n, rank))
        B = X @ X.T
        op = pkg.dense_operator(B) if hasattr(pkg, "dense_operator") else pkg.as_operator(B)
        m = min(rank + 4, n - 1)
        ws = pkg.ArnoldiWorkspace(n, m, np.float64, ctx=op.ctx)
        ws.reinitialize(0, oa.uniform_hash(5, np.arange(n)))
        for j in range(1, m + 1):
            ws.iterate_arnoldi(op, j, j)          # one step per batch: every step continues on factored columns
        H = np.array(ws.H)
        V = ws.V
        assert np.linalg.norm(V.T @ V - np.eye(m + 1)) < 200 * EPS * (m + 1), (n, rank)
        assert np.linalg.norm(B @ V[:, :m] - V @ H) < 1e-13 * max(1.0, np.linalg.norm(B)) * (m + 1), (n, rank)


@pytest.mark.parametrize("passes", ["2", "3"])
def test_host_callback_sees_unit_norm_vectors(passes, monkeypatch):
    """Inside an expansion the newest column is stored unnormalised (and, with the implicit second pass, uncorrected); a
    HOST callback -- a user's LinearMap, possibly wrapping an inner solver with absolute tolerances -- is nevertheless
    handed a vector of norm 1 (to rounding), as the reference would hand it (src/expansion.jl:106,121)."""
    monkeypatch.setenv("KS_PASSES", passes)
    A, n = _operator(np.float64, (9, 10, 11))
    A = A * 37.0                                   # sub-diagonal entries of H far from 1
    norms = []

    def mul(y, x):
        norms.append(np.linalg.norm(x))
        np.copyto(y, A @ x)

    op = pkg.host_operator(mul, n, np.float64)
    dec, hist = pkg.partialschur(op, v1=_start(np.float64, n), nev=4, which="LM", tol=1e-10, mindim=8, maxdim=16, restarts=100)
    assert hist.converged and len(norms) == hist.mvproducts
    assert np.abs(np.array(norms) - 1.0).max() < 1e-10
    ref = np.sort(np.linalg.eigvalsh(A.toarray()))[::-1][:4]
    np.testing.assert_allclose(np.sort(dec.eigenvalues.real)[::-1], ref, rtol=1e-9)


def test_column_stride_rule_is_deterministic_and_harmless(monkeypatch):
    """The leading dimension of V is padded by a FIXED rule (stride = 0xF800 mod 128 KiB for columns >= 4 MiB,
    profiles/r02_column_stride.txt) -- not by a timing search: two workspaces give bit-identical results, and the padded
    layout agrees with the unpadded one to rounding (the row ranges of the workgroups, hence the summation order, differ)."""
    m = 82                                        # 551 368 rows: 4.2 MiB columns, the rule applies
    A = laplace3d(m, m, m)
    n = A.shape[0]
    v1 = _start(np.float64, n, seed=4)
    out = []
    for rule in ("1", "1", "0"):
        monkeypatch.setenv("KS_STRIDE_RULE", rule)
        op = pkg.csr_operator(A)
        ws = pkg.ArnoldiWorkspace(n, 30, np.float64, ctx=op.ctx)
        ws.reinitialize(0, v1)
        ws.iterate_arnoldi(op, 1, 30)
        H = np.array(ws.H)
        res, orth = ws.arnoldi_relation(op, 30)
        assert res < 1e-12 * np.linalg.norm(H) * 10 and orth < np.sqrt(EPS) / 100
        out.append((H, ws.cols(0, 4)))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.linalg.norm(out[0][0] - out[2][0]) < 1e-12 * np.linalg.norm(out[2][0])
    np.testing.assert_allclose(out[0][1], out[2][1], atol=1e-10)


@pytest.mark.parametrize("scale", [1e100, 1e-100, 1e140, 1e-140])
def test_operator_scaling_within_the_documented_range(scale):
    """Norms on the device are square roots of sums of squares (no scaling pass as in BLAS nrm2), and the newest stored column
    is not normalised: k_dots scales the product by a power of two so that every accumulated quantity stays of order ||A||^2
    (DESIGN.md section 5).  Inside 1e-150 .. 1e150 the solver is scale-equivariant: eigenvalues of s A = s * eigenvalues of A,
    same number of products."""
    A, n = _operator(np.float64, (9, 10, 11))
    v1 = _start(np.float64, n, seed=8)
    kw = dict(nev=4, which="LM", tol=1e-10, mindim=8, maxdim=18, restarts=100)
    d1, h1 = pkg.partialschur(A, v1=v1, **kw)
    d2, h2 = pkg.partialschur(A * scale, v1=v1, **kw)
    assert h1.converged and h2.converged and h1.mvproducts == h2.mvproducts
    np.testing.assert_allclose(np.sort(d2.eigenvalues.real) / scale, np.sort(d1.eigenvalues.real), rtol=1e-9)
