"""Pins the ORACLE's hot path + driver (oracle/arnoldi.py) against the reference's own tests:

  KAT-1  readme.md:28-55            10 smallest eigenvalues of the 100x100 1-D Laplacian, ~174 mat-vecs
  KAT-2  test/partial_schur.jl      deterministic mat-vec counts 7 / 3 / 5
  KAT-3  test/expansion.jl:34-55    exact-zero H[5,4] + orthonormal V on an invariant subspace
  KAT-4  test/partial_schur.jl      :SR on Diagonal spectra, repeated eigenvalues
  plus   test/expansion.jl:12-32 (Arnoldi relation), test/schur_to_eigen.jl, resume via partialschur!
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import arnoldi as oa
from oracle import smalldense as sd

EPS = np.finfo(np.float64).eps
DTYPES = [np.float64, np.complex128]


def rnd(rng, dtype, *shape):
    a = rng.random(shape)
    if np.dtype(dtype).kind == "c":
        a = a + 1j * rng.random(shape)
    return a.astype(dtype)


def sprand(rng, dtype, n, density):
    M = sp.random(n, n, density=density, random_state=rng, format="csr", dtype=np.float64)
    if np.dtype(dtype).kind == "c":
        Mi = sp.random(n, n, density=density, random_state=rng, format="csr", dtype=np.float64)
        M = (M + 1j * Mi).tocsr()
    return M.astype(dtype)


# ------------------------------------------------------------------ rng
def test_uniform_hash_is_a_pure_function_and_uniform():
    u = oa.uniform_hash(oa.DEFAULT_SEED, np.arange(100000))
    assert (u >= 0).all() and (u < 1).all()
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 2e-3
    # partition independence: rows 1000.. generated on another "rank"
    v = np.empty(500)
    oa.rand_fill(v, oa.DEFAULT_SEED, row_offset=1000)
    assert (v == u[1000:1500]).all()
    # LITERAL golden values, obtained independently of oracle/arnoldi.py (pure-Python integer arithmetic; VERDICT r2: the
    # previous constant was computed from the function it checked):
    #   * the published known answer of the splitmix64 generator (Vigna's splitmix64.c): state 1234567 -> first output
    #     6457827717110365317 -- our finaliser is exactly one step of it, splitmix64(x) = mix(x + 0x9E3779B97F4A7C15);
    #   * u[i] = (splitmix64(seed xor i) >> 11) * 2^-53 for seed 20240917, i = 0..3, as exact hexadecimal doubles.
    assert int(oa.splitmix64(np.uint64(1234567))) == 6457827717110365317
    g = oa.uniform_hash(20240917, np.arange(4))
    assert [float(x).hex() for x in g] == GOLDEN_HASH_HEX
    assert g.tolist() == [0.9248160014551147, 0.418340663264014, 0.11549796239421439, 0.27945786907302395]


GOLDEN_HASH_HEX = ["0x1.d9817ba22268dp-1", "0x1.ac617ead393aep-2", "0x1.d9146433cdfb8p-4", "0x1.1e2a34211d2bep-2"]


# ------------------------------------------------------------------ test/expansion.jl
def test_initialization():
    ws = oa.ArnoldiWorkspace.from_dims(np.float64, 5, 3)
    oa.reinitialize(ws)
    assert np.linalg.norm(ws.V[:, 0]) == pytest.approx(1.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_arnoldi_factorization(dtype):
    rng = np.random.default_rng(5)
    n, mx = 10, 6
    A = (sprand(rng, dtype, n, 0.1) + sp.identity(n)).tocsr()
    ws = oa.ArnoldiWorkspace.from_dims(dtype, n, mx)
    oa.reinitialize(ws)
    V, H = ws.V, ws.H
    oa.iterate_arnoldi(A, ws, 1, 3)
    np.testing.assert_allclose(A @ V[:, :3], V[:, :4] @ H[:4, :3], atol=1e-13)
    assert np.linalg.norm(V[:, :4].conj().T @ V[:, :4] - np.eye(4)) < np.sqrt(EPS) / 100
    oa.iterate_arnoldi(A, ws, 4, mx)
    np.testing.assert_allclose(A @ V[:, :mx], V @ H, atol=1e-13)
    assert np.linalg.norm(V.conj().T @ V - np.eye(mx + 1)) < np.sqrt(EPS) / 100


def test_invariant_subspace():  # KAT-3
    rng = np.random.default_rng(6)
    A = np.zeros((8, 8))
    A[:4, :4] = rng.random((4, 4))
    A[4:, 4:] = rng.random((4, 4))
    ws = oa.ArnoldiWorkspace.from_dims(np.float64, 8, 5)
    ws.V[:, 0] = 0
    ws.V[0, 0] = 1
    st = {}
    oa.iterate_arnoldi(A, ws, 1, 5, st)
    assert np.linalg.norm(ws.V.T @ ws.V - np.eye(6)) < np.sqrt(EPS) / 100
    assert ws.H[4, 3] == 0  # jl: iszero(H[5, 4])
    assert st["breakdowns"] == 1


# ------------------------------------------------------------------ KAT-1
def laplace1d(n):
    return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")


def test_readme_example():
    A = laplace1d(100)
    dec, hist = oa.partialschur(A, nev=10, tol=1e-6, which="SR")
    assert hist.converged and hist.nconverged == 10
    # readme.md:51-52 says 174 with Julia's random v1; "+- a few" for another start vector
    assert 150 <= hist.mvproducts <= 200
    printed = [
        0.0009674354160236865, 0.003868805732811139, 0.008701304061962657, 0.01546025527344699,
        0.024139120518486677, 0.0347295035554728, 0.04722115887278571, 0.06160200160067088,
        0.0778581192025522, 0.09597378493453936,
    ]  # readme.md:40-49
    np.testing.assert_allclose(dec.eigenvalues.real, printed, rtol=0, atol=1e-6 * 0.1)
    exact = 2 - 2 * np.cos(np.arange(1, 11) * np.pi / 101)
    np.testing.assert_allclose(dec.eigenvalues.real, exact, atol=1e-7)
    assert (dec.eigenvalues.imag == 0).all()
    res = np.linalg.norm(A @ dec.Q - dec.Q @ dec.R)
    assert res < 1e-6  # readme.md:54-55 prints 6.39e-8
    vals, vecs = oa.partialeigen(dec)
    assert np.linalg.norm(A @ vecs - vecs * vals) < 1e-6
    assert str(hist).startswith("Converged: 10 of 10 eigenvalues in ")


# ------------------------------------------------------------------ test/partial_schur.jl
@pytest.mark.parametrize("dtype", DTYPES)
def test_low_rank(dtype):  # KAT-2: mvproducts == 7
    rng = np.random.default_rng(7)
    A = rnd(rng, dtype, 10, 3)
    B = A @ A.conj().T
    dec, hist = oa.partialschur(B, nev=5, mindim=5, maxdim=7, tol=EPS)
    assert hist.converged
    assert hist.mvproducts == 7
    assert np.linalg.norm(dec.Q.conj().T @ dec.Q - np.eye(dec.Q.shape[1])) < 1000 * EPS
    assert np.linalg.norm(B @ dec.Q - dec.Q @ dec.R) < 1000 * EPS * max(1.0, np.linalg.norm(B))
    assert np.linalg.norm(np.diag(dec.R)[3:5]) < 1000 * EPS * max(1.0, np.linalg.norm(B))


def test_right_number_type():
    rng = np.random.default_rng(8)
    A = (rng.random((10, 10)) < 0.5).astype(np.int64)
    assert oa.vtype(A) == np.float64
    dec, hist = oa.partialschur(A, nev=2, mindim=3, maxdim=8)
    assert dec.Q.dtype == np.float64


def test_all_eigenvalues_small_matrix():  # KAT-2: mvproducts == 3
    rng = np.random.default_rng(9)
    A = rng.random((3, 3))
    dec, hist = oa.partialschur(A)
    assert hist.converged and hist.mvproducts == 3


def test_incorrect_input():
    rng = np.random.default_rng(10)
    A = rng.random((6, 6))
    with pytest.raises(oa.DimensionMismatch):
        oa.partialschur(rng.random((4, 3)))
    with pytest.raises(oa.ArgumentError):
        oa.partialschur(A, mindim=5, maxdim=3)
    with pytest.raises(oa.ArgumentError):
        oa.partialschur(A, nev=5, mindim=3)
    with pytest.raises(oa.ArgumentError):
        oa.partialschur(A, nev=5, maxdim=3)
    with pytest.raises(oa.ArgumentError):
        oa.partialschur(A, nev=10)
    with pytest.raises(oa.ArgumentError):
        oa.partialschur(A, nev=0)
    with pytest.raises(oa.ArgumentError):
        oa.partialschur(A, which="XX")
    with pytest.raises(oa.ArgumentError):
        oa.partialschur(A, v1=np.ones(5))


def test_eigenvector_as_initial_vector():
    rng = np.random.default_rng(11)
    A = rng.random((30, 30))
    A = A + A.T
    lams, X = np.linalg.eigh(A)
    lam, x = lams[-1], X[:, -1]
    x0 = x.copy()
    dec, hist = oa.partialschur(A, v1=x, nev=2, tol=1e-8)
    assert (x == x0).all()  # v1 is never mutated (src/run.jl:38)
    assert hist.converged
    assert np.linalg.norm(A @ dec.Q - dec.Q @ dec.R) < 1e-7
    assert abs(dec.eigenvalues.real.max() - lam) < 1e-7


def test_target_non_dominant():  # KAT-4
    d = np.concatenate([np.arange(1, 10.0001, 0.1), [50, 51, 52, 53]])
    A = sp.diags(d).tocsr()
    dec, hist = oa.partialschur(A, which="SR")
    assert (sd.eigenvalues(np.asfortranarray(dec.R)).real <= 10).all()


def test_repeated_eigenvalues():  # KAT-4
    d = np.concatenate([np.arange(1, 9.0001, 0.1), [9.97, 9.98, 9.99, 10.0, 10.0, 10.0]])
    A = sp.diags(d).tocsr()
    dec, hist = oa.partialschur(A, nev=5, maxdim=20, tol=1e-12)
    assert hist.converged
    k = dec.Q.shape[1]
    assert np.linalg.norm(dec.Q.T @ dec.Q - np.eye(k)) < 100 * EPS
    assert np.linalg.norm(A @ dec.Q - dec.Q @ dec.R) < A.shape[0] * 1e-12


@pytest.mark.parametrize("dtype", DTYPES)
def test_zero_matrix(dtype):  # KAT-2: mvproducts == nconverged == 5, residual exactly 0
    A = np.zeros((5, 5), dtype=dtype)
    dec, hist = oa.partialschur(A)
    assert hist.converged
    assert hist.mvproducts == hist.nconverged == 5
    assert np.linalg.norm(dec.Q.conj().T @ dec.Q - np.eye(5)) < 100 * EPS
    assert np.linalg.norm(A @ dec.Q - dec.Q @ dec.R) == 0


def test_passing_initial_schur_decomp():
    rng = np.random.default_rng(12)
    A = rng.random((100, 100))
    V = np.asfortranarray(rng.random((100, 21)))
    H = np.asfortranarray(rng.random((21, 20)))
    ws = oa.ArnoldiWorkspace(V, H)
    F, hist = oa.partialschur_(A, ws, nev=3, tol=1e-12)
    assert hist.converged and hist.nconverged in (3, 4)
    assert np.linalg.norm(A @ F.Q - F.Q @ F.R) < 1e-10
    assert np.shares_memory(F.Q, ws.V) and np.shares_memory(F.R, ws.H)  # views, src/run.jl:149-150
    F, hist = oa.partialschur_(A, ws, nev=5, start_from=hist.nconverged + 1, tol=1e-8)
    assert hist.converged and hist.nconverged in (5, 6)
    assert np.linalg.norm(A @ F.Q - F.Q @ F.R) < 1e-6


def test_partialschur_bang_argument_checks():
    rng = np.random.default_rng(13)
    A = rng.random((50, 50))
    ws = oa.ArnoldiWorkspace.from_dims(np.float64, 50, 20)
    with pytest.raises(oa.ArgumentError):
        oa.partialschur_(A, ws, maxdim=21)
    with pytest.raises(oa.ArgumentError):
        oa.partialschur_(A, ws, start_from=0)
    with pytest.raises(oa.ArgumentError):
        oa.partialschur_(A, ws, start_from=21)
    with pytest.raises(oa.ArgumentError):
        oa.ArnoldiWorkspace(np.zeros((50, 21)), np.zeros((20, 20)))
    with pytest.raises(oa.ArgumentError):
        oa.ArnoldiWorkspace(np.zeros((50, 21)), np.zeros((21, 21)))
    with pytest.raises(oa.ArgumentError):
        oa.ArnoldiWorkspace.from_dims(np.float64, 5, 6)


# ------------------------------------------------------------------ test/schur_to_eigen.jl
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("seed", range(1, 11))
def test_schur_to_eigen(dtype, seed):
    rng = np.random.default_rng(seed)
    A = (sp.diags(np.arange(1, 101, dtype=float)) + sprand(rng, dtype, 100, 0.01)).tocsr()
    eps_ = np.sqrt(EPS)
    dec, hist = oa.partialschur(A, nev=10, tol=eps_, restarts=200, seed=seed)
    assert hist.converged
    vals, vecs = oa.partialeigen(dec)
    for i in range(10):
        assert np.linalg.norm(A @ vecs[:, i] - vecs[:, i] * vals[i]) < eps_ * abs(vals[i])


# ------------------------------------------------------------------ beyond the reference tests
def test_nonsymmetric_complex_pairs_lm():
    """Real matrix with planted complex-conjugate outliers: exercises 2x2 blocks through
    the Schur restart, the never-split-a-pair rule (src/run.jl:298,321) and swap12/21/22."""
    rng = np.random.default_rng(14)
    n = 200
    A = sp.random(n, n, density=0.02, random_state=rng, format="lil") * 0.2
    A = A.tolil()
    blocks = [(5.0, 3.0), (4.0, -2.5), (-6.0, 1.0)]
    for b, (a, c) in enumerate(blocks):
        i = 2 * b
        A[i, i] = a
        A[i + 1, i + 1] = a
        A[i, i + 1] = c
        A[i + 1, i] = -c
    A[6, 6] = 7.5
    A = A.tocsr()
    dec, hist = oa.partialschur(A, nev=6, which="LM", tol=1e-10)
    assert hist.converged
    k = dec.Q.shape[1]
    assert k in (6, 7)
    assert np.linalg.norm(A @ dec.Q - dec.Q @ dec.R) < 1e-8
    assert np.linalg.norm(dec.Q.T @ dec.Q - np.eye(k)) < 1e-12
    ref = np.linalg.eigvals(A.toarray())
    ref = ref[np.argsort(-np.abs(ref))][:k]
    got = dec.eigenvalues[np.argsort(-np.abs(dec.eigenvalues))]
    np.testing.assert_allclose(np.sort_complex(got), np.sort_complex(ref), atol=1e-7)
    # pairs: +imag listed first (src/eigvals.jl:20-24)
    for i in range(k - 1):
        if dec.R[i + 1, i] != 0:
            assert dec.eigenvalues[i].imag > 0 and dec.eigenvalues[i + 1] == np.conj(dec.eigenvalues[i])


@pytest.mark.parametrize("which", ["LM", "LR", "SR", "LI", "SI"])
def test_all_targets_complex(which):
    rng = np.random.default_rng(15)
    n = 60
    d = rng.standard_normal(n) * 3 + 1j * rng.standard_normal(n) * 3
    A = (sp.diags(d) + sprand(rng, np.complex128, n, 0.02) * 0.01).tocsr()
    dec, hist = oa.partialschur(A, nev=4, which=which, tol=1e-10)
    assert hist.converged
    ref = np.linalg.eigvals(A.toarray())
    key = {"LM": lambda z: -abs(z), "LR": lambda z: -z.real, "SR": lambda z: z.real, "LI": lambda z: -z.imag, "SI": lambda z: z.imag}[which]
    want = sorted(ref, key=key)[:4]
    got = sorted(dec.eigenvalues[:4], key=key)
    np.testing.assert_allclose(got, want, atol=1e-7)
    assert np.linalg.norm(A @ dec.Q - dec.Q @ dec.R) < 1e-8


def test_laplace3d_anisotropic_parity_size():
    """Analytic KAT: 3-D 7-point Laplacian on an anisotropic grid (no degenerate eigenvalues)."""
    from oracle.matrices import laplace3d, laplace3d_eigs

    mx, my, mz = 8, 9, 10
    A = laplace3d(mx, my, mz)
    dec, hist = oa.partialschur(A, nev=6, which="SR", tol=1e-10, maxdim=30)
    assert hist.converged
    exact = laplace3d_eigs(mx, my, mz)[:6]
    np.testing.assert_allclose(np.sort(dec.eigenvalues.real)[:6], exact, atol=1e-8)
    assert np.linalg.norm(A @ dec.Q - dec.Q @ dec.R) < 1e-8
