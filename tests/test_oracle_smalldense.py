"""Pins the ORACLE's small dense host kernels (oracle/smalldense.py, oracle/givens.py) against the
reference's own unit tests -- KAT-5 of SURVEY.md section 8c -- replayed with a seeded RNG:

  test/givens_rotation.jl, test/ordering.jl, test/schurfact.jl, test/sylvester.jl,
  test/sort_schur.jl, test/collect_eigen.jl, test/householder.jl.

The explicit matrices (hard QR cases, Stewart, Bai-Demmel, identical eigenvalues,
[1 -1/4; 1 2]) are DATA taken from those tests.
"""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle import smalldense as sd
from oracle.givens import givens_complex, givens_real

EPS = np.finfo(np.float64).eps
DTYPES = [np.float64, np.complex128]


def rnd(rng, dtype, *shape):
    a = rng.random(shape)
    if np.dtype(dtype).kind == "c":
        a = a + 1j * rng.random(shape)
    return np.asfortranarray(a.astype(dtype))


def rndn(rng, dtype, *shape):
    a = rng.standard_normal(shape)
    if np.dtype(dtype).kind == "c":
        a = (a + 1j * rng.standard_normal(shape)) / np.sqrt(2)
    return np.asfortranarray(a.astype(dtype))


def reim_sorted(x):
    x = np.asarray(x, dtype=np.complex128)
    return x[np.lexsort((x.imag, x.real))]


# ------------------------------------------------------------------ givensAlgorithm
def test_givens_real_zeroes_second_component():
    rng = np.random.default_rng(1)
    for _ in range(200):
        f, g = rng.standard_normal(2) * 10.0 ** rng.integers(-8, 8)
        c, s, r = givens_real(f, g)
        assert abs(c * c + s * s - 1) < 4 * EPS
        assert abs(c * f + s * g - r) <= 4 * EPS * abs(r)
        assert abs(-s * f + c * g) <= 4 * EPS * abs(r)
    assert givens_real(3.0, 0.0) == (1.0, 0.0, 3.0)
    assert givens_real(0.0, 2.0) == (0.0, 1.0, 2.0)
    c, s, r = givens_real(-2.0, 1.0)  # |f| > |g|  =>  c >= 0 (dlartg sign convention)
    assert c > 0 and r < 0


def test_givens_real_extreme_scales():
    for f, g in [(1e300, 1e300), (1e-300, 3e-300), (1e200, 1e-200)]:
        c, s, r = givens_real(f, g)
        assert np.isfinite([c, s, r]).all()
        assert abs(c * c + s * s - 1) < 8 * EPS
        assert abs(r) == pytest.approx(np.hypot(f, g), rel=1e-14)


def test_givens_complex():
    rng = np.random.default_rng(2)
    for _ in range(200):
        f = complex(*rng.standard_normal(2))
        g = complex(*rng.standard_normal(2))
        c, s, r = givens_complex(f, g)
        assert isinstance(c, float) and c >= 0
        assert abs(c * c + abs(s) ** 2 - 1) < 8 * EPS
        assert abs(c * f + s * g - r) <= 8 * EPS * abs(r)
        assert abs(-np.conj(s) * f + c * g) <= 8 * EPS * abs(r)
    c, s, r = givens_complex(0j, 3 + 4j)
    assert c == 0.0 and abs(r - 5.0) < 1e-15 and abs(s * (3 + 4j) - r) < 1e-15
    c, s, r = givens_complex(2 + 1j, 0j)
    assert (c, s, r) == (1.0, 0j, 2 + 1j)


# ------------------------------------------------------------------ test/givens_rotation.jl
@pytest.mark.parametrize("dtype", DTYPES)
def test_rotation2_lmul_rmul(dtype):
    rng = np.random.default_rng(3)
    A = rnd(rng, dtype, 6, 5)
    G = sd.Rotation2(rng.random(), rnd(rng, dtype, 1)[0], 1)
    Gm = G.matrix(6, dtype)
    B = A.copy()
    sd.lmul(G, B, 1, 3)
    np.testing.assert_allclose(B, np.hstack([A[:, :1], Gm @ A[:, 1:4], A[:, 4:]]), rtol=1e-14)
    B = A.copy()
    sd.lmul(G, B)
    np.testing.assert_allclose(B, Gm @ A, rtol=1e-14)

    A = rnd(rng, dtype, 10, 5)
    Gm = G.matrix(5, dtype)
    B = A.copy()
    sd.rmul(B, G, 1, 3)  # rmul! multiplies by G' (test/givens_rotation.jl:25-26)
    np.testing.assert_allclose(B, np.vstack([A[:1], A[1:4] @ Gm.conj().T, A[4:]]), rtol=1e-14)
    B = A.copy()
    sd.rmul(B, G)
    np.testing.assert_allclose(B, A @ Gm.conj().T, rtol=1e-14)


@pytest.mark.parametrize("dtype", DTYPES)
def test_rotation3_lmul_rmul(dtype):
    rng = np.random.default_rng(4)
    A = rnd(rng, dtype, 6, 5)
    G = sd.Rotation3(rng.random(), rnd(rng, dtype, 1)[0], rng.random(), rnd(rng, dtype, 1)[0], 1)
    Gm = G.matrix(6, dtype)
    B = A.copy()
    sd.lmul(G, B, 1, 3)
    np.testing.assert_allclose(B, np.hstack([A[:, :1], Gm @ A[:, 1:4], A[:, 4:]]), rtol=1e-14)
    B = A.copy()
    sd.lmul(G, B)
    np.testing.assert_allclose(B, Gm @ A, rtol=1e-14)
    A = rnd(rng, dtype, 10, 5)
    Gm = G.matrix(5, dtype)
    B = A.copy()
    sd.rmul(B, G, 1, 3)
    np.testing.assert_allclose(B, np.vstack([A[:1], A[1:4] @ Gm.conj().T, A[4:]]), rtol=1e-14)
    B = A.copy()
    sd.rmul(B, G)
    np.testing.assert_allclose(B, A @ Gm.conj().T, rtol=1e-14)


# ------------------------------------------------------------------ test/ordering.jl
def test_stable_permutation_ordering():
    xs = np.array([1 + 3j, 1 - 3j, 4 + 0j])
    for which in ("SR",):  # Forward with f = real
        assert list(sd.sort_perm(np.arange(3), xs, sd.get_order(which))) == [0, 1, 2]
    fwd_abs = lambda a, b: sd._isless(abs(a), abs(b))  # OrderBy(abs)
    assert list(sd.sort_perm(np.arange(3), xs, fwd_abs)) == [0, 1, 2]
    for which in ("LR", "LM"):  # Backward with f = real, abs  ->  [3, 1, 2] (1-based)
        assert list(sd.sort_perm(np.arange(3), xs, sd.get_order(which))) == [2, 0, 1]


def test_isless_total_order():
    assert sd._isless(-0.0, 0.0) and not sd._isless(0.0, -0.0)
    assert sd._isless(1.0, float("nan")) and not sd._isless(float("nan"), 1.0)
    with pytest.raises(ValueError):
        sd.get_order("XX")


# ------------------------------------------------------------------ test/schurfact.jl
@pytest.mark.parametrize(
    "H0,zero21",
    [
        (np.array([[1.0, 2.0], [3.0, 4.0]]), True),
        (np.array([[1.0, 2.0], [0.0, 4.0]]), True),
        (np.array([[1.0, 4.0], [-5.0, 3.0]]), False),
    ],
)
def test_schurfact_2x2(H0, zero21):
    H = np.asfortranarray(H0.copy())
    Q = np.eye(2, order="F")
    assert sd.local_schurfact(H, 0, 1, Q, EPS, 2)
    assert np.linalg.norm(H0 @ Q - Q @ H) < 10 * EPS
    np.testing.assert_allclose(reim_sorted(sd.eigenvalues(H)), reim_sorted(np.linalg.eigvals(H0)), rtol=1e-13)
    if zero21:
        assert H[1, 0] == 0


def normal_hessenberg(rng, dtype, vals):
    """test/utils.jl:8-33."""
    n = len(vals)
    Qm, _ = np.linalg.qr(rndn(rng, dtype, n, n))
    vals = np.asarray(vals)
    if np.dtype(dtype).kind == "f" and vals.dtype.kind == "c":
        D = np.zeros((n, n))
        i = 0
        while i < n:
            if vals[i].imag != 0:
                D[i, i] = vals[i].real
                D[i + 1, i] = vals[i].imag
                D[i, i + 1] = -vals[i].imag
                D[i + 1, i + 1] = vals[i].real
                i += 2
            else:
                D[i, i] = vals[i].real
                i += 1
        A = Qm @ D @ Qm.T
    else:
        A = Qm @ np.diag(vals) @ Qm.conj().T
    return np.triu(sla.hessenberg(A), -1)


def is_hessenberg(H):
    return np.linalg.norm(np.tril(H, -2)) == 0


@pytest.mark.parametrize("i", range(5))
def test_schurfact_real_partial_block(i):
    rng = np.random.default_rng(100 + i)
    n = 10
    Q = np.eye(n, order="F")
    H = np.asfortranarray(np.triu(rng.standard_normal((n, n))))
    H[i : n - i, i : n - i] = normal_hessenberg(rng, np.float64, np.arange(i + 1, n - i + 1, dtype=float))
    Hp = H.copy(order="F")
    assert sd.local_schurfact(Hp, i, n - i - 1, Q)
    for j in range(i, n - i - 1):
        t = Hp[j, j] + Hp[j + 1, j + 1]
        d = Hp[j, j] * Hp[j + 1, j + 1] - Hp[j + 1, j] * Hp[j, j + 1]
        assert sd.is_offdiagonal_small(Hp, j) or t * t < 4 * d
    assert is_hessenberg(Hp)
    assert np.linalg.norm(H @ Q - Q @ Hp) < 1000 * EPS
    np.testing.assert_allclose(reim_sorted(np.linalg.eigvals(H)), reim_sorted(np.linalg.eigvals(Hp)), atol=1e-10)


@pytest.mark.parametrize("i", range(5))
def test_schurfact_real_conjugate_pairs(i):
    """Real path with complex-conjugate eigenvalues (double shift), cf. test/utils.jl:17-33."""
    rng = np.random.default_rng(150 + i)
    n = 10
    vals = np.array([1 + 2j, 1 - 2j, 3, -1 + 0.5j, -1 - 0.5j, 4, 5 + 1j, 5 - 1j, -2, 0.5], dtype=complex)
    H = np.asfortranarray(normal_hessenberg(rng, np.float64, vals))
    Hp = H.copy(order="F")
    Q = np.eye(n, order="F")
    assert sd.local_schurfact(Hp, 0, n - 1, Q)
    assert is_hessenberg(Hp)
    assert np.linalg.norm(H @ Q - Q @ Hp) < 1000 * EPS
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 100 * EPS
    np.testing.assert_allclose(reim_sorted(sd.eigenvalues(Hp)), reim_sorted(vals), atol=1e-10)


@pytest.mark.parametrize("i", range(5))
def test_schurfact_complex_partial_block(i):
    rng = np.random.default_rng(200 + i)
    n = 10
    Q = np.eye(n, dtype=np.complex128, order="F")
    H = np.asfortranarray(np.triu(rndn(rng, np.complex128, n, n)))
    H[i : n - i, i : n - i] = normal_hessenberg(rng, np.complex128, np.arange(i + 1, n - i + 1) * (1 + 1j))
    Hp = H.copy(order="F")
    assert sd.local_schurfact(Hp, i, n - i - 1, Q)
    for j in range(i, n - i - 1):
        assert Hp[j + 1, j] == 0
    assert is_hessenberg(Hp)
    assert np.linalg.norm(H @ Q - Q @ Hp) < 1000 * EPS
    np.testing.assert_allclose(reim_sorted(np.linalg.eigvals(H)), reim_sorted(np.linalg.eigvals(Hp)), atol=1e-10)


def test_schurfact_nearly_repeated():  # test/schurfact.jl:123-135
    e = EPS
    M = np.asfortranarray(np.array([[2, 0, 0], [5 * e, 1 - e, 2 * e], [0, 3 * e, 1 + e]], dtype=float))
    assert sd.local_schurfact(M)


def test_schurfact_in_the_wild():  # test/schurfact.jl:137-158
    H1 = np.asfortranarray(
        np.array(
            [
                [-9.000000046596169, 9.363971416904122e-6, 0.6216202324428521, 0.783119615978767],
                [-3.1249216068055166e-10, -9.000000125049475, -0.005030734831215954, 0.026538692060151765],
                [0.0, 2.5838932886290116e-12, -8.999999884550379, -4.118678562647915e-7],
                [0.0, 0.0, 5.499735555858365e-9, -8.99999994380397],
            ]
        )
    )
    assert sd.local_schurfact(H1)
    H2 = np.asfortranarray(
        np.array(
            [
                [-9.99999999890572, -5.359512176950441e-5, 0.5057150345932383],
                [6.673511665530937e-11, -9.999999865827567, -0.0009029114103036593],
                [0.0, 1.432733142195386e-11, -10.000000096783797],
            ]
        )
    )
    assert sd.local_schurfact(H2)


def test_exactly_repeated_2x2():  # test/schurfact.jl:160-174
    A = np.array([[1.0, -0.25], [1.0, 2.0]])
    is_real, c, s = sd.upper_triangular_2x2(A[0, 0], A[0, 1], A[1, 0], A[1, 1])
    assert is_real
    G = np.array([[c, s], [-s, c]])
    np.testing.assert_allclose(G @ A @ G.T, np.array([[1.5, -1.25], [0, 1.5]]), atol=1e-15)
    np.testing.assert_allclose(G.T @ G, np.eye(2), atol=1e-15)
    is_single, lam = sd.use_single_shift(A[0, 0], A[0, 1], A[1, 0], A[1, 1])
    assert is_single and lam == pytest.approx(1.5)


# ------------------------------------------------------------------ test/sylvester.jl
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("p,q", [(2, 2), (2, 1), (1, 2)])
def test_tiny_sylvester(dtype, p, q):
    rng = np.random.default_rng(7)
    A, B, C = rnd(rng, dtype, p, p), rnd(rng, dtype, q, q), rnd(rng, dtype, p, q)
    X, singular = sd.sylv(A, B, C)
    np.testing.assert_allclose(A @ X - X @ B, C, rtol=1e-10)
    assert not singular


@pytest.mark.parametrize("dtype", DTYPES)
def test_singular_sylvester(dtype):
    rng = np.random.default_rng(8)
    T = lambda x: np.array(x, dtype=dtype)
    assert sd.sylv(T([[1, 2], [0, 1]]), T([[1, 3], [0, 1]]), rnd(rng, dtype, 2, 2))[1]
    assert sd.sylv(T([[1]]), T([[1, 3], [0, 1]]), rnd(rng, dtype, 1, 2))[1]
    assert sd.sylv(T([[1, 2], [0, 1]]), T([[1]]), rnd(rng, dtype, 2, 1))[1]


# ------------------------------------------------------------------ test/sort_schur.jl
@pytest.mark.parametrize("dtype", DTYPES)
def test_swap11(dtype):
    rng = np.random.default_rng(9)
    R1 = np.asfortranarray(np.triu(rnd(rng, dtype, 2, 2)))
    R2, Q2 = R1.copy(order="F"), np.eye(2, dtype=dtype, order="F")
    sd.swap11(R2, 0, Q2)
    assert R2[0, 0] == pytest.approx(R1[1, 1]) and R1[0, 0] == pytest.approx(R2[1, 1])
    np.testing.assert_allclose(R1 @ Q2, Q2 @ R2, atol=1e-14)


@pytest.mark.parametrize("dtype", DTYPES)
def test_swap12(dtype):
    rng = np.random.default_rng(10)
    R1 = np.asfortranarray(np.triu(rnd(rng, dtype, 3, 3)))
    R1[2, 1] = rnd(rng, dtype, 1)[0]
    R2, Q2 = R1.copy(order="F"), np.eye(3, dtype=dtype, order="F")
    sd.swap12(R2, 0, Q2)
    assert R2[2, 0] == 0 and R2[2, 1] == 0
    assert R1[0, 0] == pytest.approx(R2[2, 2])
    np.testing.assert_allclose(reim_sorted(np.linalg.eigvals(R1[1:, 1:])), reim_sorted(np.linalg.eigvals(R2[:2, :2])), atol=1e-13)
    np.testing.assert_allclose(R1 @ Q2, Q2 @ R2, atol=1e-13)


@pytest.mark.parametrize("dtype", DTYPES)
def test_swap21(dtype):
    rng = np.random.default_rng(11)
    R1 = np.asfortranarray(np.triu(rnd(rng, dtype, 3, 3)))
    R1[1, 0] = rnd(rng, dtype, 1)[0]
    R2, Q2 = R1.copy(order="F"), np.eye(3, dtype=dtype, order="F")
    sd.swap21(R2, 0, Q2)
    assert R2[1, 0] == 0 and R2[2, 0] == 0
    assert R1[2, 2] == pytest.approx(R2[0, 0])
    np.testing.assert_allclose(reim_sorted(np.linalg.eigvals(R1[:2, :2])), reim_sorted(np.linalg.eigvals(R2[1:, 1:])), atol=1e-13)
    np.testing.assert_allclose(R1 @ Q2, Q2 @ R2, atol=1e-13)


@pytest.mark.parametrize("dtype", DTYPES)
def test_swap22(dtype):
    rng = np.random.default_rng(12)
    R1 = np.asfortranarray(np.triu(rnd(rng, dtype, 4, 4)))
    R1[1, 0] = rnd(rng, dtype, 1)[0]
    R1[3, 2] = rnd(rng, dtype, 1)[0]
    R2, Q2 = R1.copy(order="F"), np.eye(4, dtype=dtype, order="F")
    sd.swap22(R2, 0, Q2)
    assert R2[2, 0] == 0 and R2[3, 0] == 0 and R2[2, 1] == 0 and R2[3, 1] == 0
    np.testing.assert_allclose(reim_sorted(np.linalg.eigvals(R1[:2, :2])), reim_sorted(np.linalg.eigvals(R2[2:, 2:])), atol=1e-12)
    np.testing.assert_allclose(reim_sorted(np.linalg.eigvals(R1[2:, 2:])), reim_sorted(np.linalg.eigvals(R2[:2, :2])), atol=1e-12)
    np.testing.assert_allclose(R1 @ Q2, Q2 @ R2, atol=1e-12)


def _rot_check(R, Ra, Q):
    op1 = lambda M: np.linalg.norm(M, 1)
    assert op1(R - Q @ Ra @ Q.conj().T) < 10 * EPS * op1(R) * 10
    assert np.linalg.norm(Q.conj().T @ Q - np.eye(Q.shape[0])) < 10 * EPS * 10


@pytest.mark.parametrize("dtype", DTYPES)
def test_rotate_right_single(dtype):  # test/sort_schur.jl:113-140
    rng = np.random.default_rng(13)
    R = np.asfortranarray(np.triu(rnd(rng, dtype, 10, 10)))
    Q = np.eye(10, dtype=dtype, order="F")
    R[3, 4] = -2
    R[4, 3] = 2
    before = sd.eigenvalues(R)
    Ra = R.copy(order="F")
    sd.rotate_right(Ra, 0, 9, Q)
    after = sd.eigenvalues(Ra)
    _rot_check(R, Ra, Q)
    idx = np.arange(10)
    for i, j in zip(idx, np.roll(idx, 1)):  # circshift(1:10, -1)
        assert before[i] == pytest.approx(after[(i + 1) % 10], rel=1e-10, abs=1e-12)


@pytest.mark.parametrize("dtype", DTYPES)
def test_rotate_right_two_pairs(dtype):  # test/sort_schur.jl:142-178
    rng = np.random.default_rng(14)
    R = np.asfortranarray(np.triu(rnd(rng, dtype, 10, 10)))
    Q = np.eye(10, dtype=dtype, order="F")
    R[2, 1] = -2
    R[1, 2] = 2
    R[6, 5] = 3
    R[5, 6] = -2
    before = sd.eigenvalues(R)
    Ra = R.copy(order="F")
    sd.rotate_right(Ra, 2, 5, Q)  # jl: rotate_right!(R, 3, 6, Q)
    after = sd.eigenvalues(Ra)
    _rot_check(R, Ra, Q)
    assert before[0] == after[0]
    src = list(range(1, 7))
    dst = src[2:] + src[:2]  # circshift(2:7, -2)
    for i, j in zip(src, dst):
        assert before[i] == pytest.approx(after[j], rel=1e-10, abs=1e-12)
    assert (before[7:] == after[7:]).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_rotate_right_block_on_right(dtype):  # test/sort_schur.jl:180-213
    rng = np.random.default_rng(15)
    R = np.asfortranarray(np.triu(rnd(rng, dtype, 10, 10)))
    Q = np.eye(10, dtype=dtype, order="F")
    R[5, 6] = -2
    R[6, 5] = 2
    before = sd.eigenvalues(R)
    Ra = R.copy(order="F")
    sd.rotate_right(Ra, 1, 5, Q)
    after = sd.eigenvalues(Ra)
    _rot_check(R, Ra, Q)
    assert before[0] == after[0]
    src = list(range(1, 7))
    dst = src[2:] + src[:2]
    for i, j in zip(src, dst):
        assert before[i] == pytest.approx(after[j], rel=1e-10, abs=1e-12)
    assert (before[7:] == after[7:]).all()


def stewart(tau):
    return np.asfortranarray(
        np.array(
            [
                [7 + 1 / 1000, -87, (39 + 2 / 5) * tau, (22 + 2 / 5) * tau],
                [5, 7 + 1 / 1000, -(12 + 2 / 5) * tau, 36 * tau],
                [0, 0, 7 + 1 / 100, -7567 / 10000],
                [0, 0, 37, 7 + 1 / 100],
            ],
            dtype=float,
        )
    )


@pytest.mark.parametrize("tau", [1, 10, 100])
def test_stewart_example(tau):  # test/sort_schur.jl:256-278
    B = stewart(float(tau))
    before = sd.eigenvalues(B)
    sd.swap22(B, 0)
    after = sd.eigenvalues(B)
    assert abs(before[0]) == pytest.approx(abs(after[2]), rel=1e-8)
    assert abs(before[2]) == pytest.approx(abs(after[0]), rel=1e-8)


def test_small_eigenvalue_separation():  # test/sort_schur.jl:281-301 (Bai & Demmel)
    A = np.asfortranarray(
        np.array(
            [[1, -100, 400, -1000], [1 / 100, 1, 1200, -10], [0, 0, 1 + EPS, -1 / 100], [0, 0, 100, 1 + EPS]],
            dtype=float,
        )
    )
    Ap = A.copy(order="F")
    Q = np.eye(4, order="F")
    sd.swap22(Ap, 0, Q)
    op1 = lambda M: np.linalg.norm(M, 1)
    assert op1(np.eye(4) - Q.T @ Q) < 10 * EPS
    assert op1(A @ Q - Q @ Ap) < op1(A) * EPS * 4
    before, after = sd.eigenvalues(A), sd.eigenvalues(Ap)
    assert abs(before[0]) == pytest.approx(abs(after[2]), rel=1e-6)
    assert abs(before[2]) == pytest.approx(abs(after[0]), rel=1e-6)


def test_identical_eigenvalues_no_blowup():  # test/sort_schur.jl:303-320
    A = np.asfortranarray(np.array([[1, 2, 3, 4], [0, 1, 5, 6], [0, 0, 1, 7], [0, 0, 0, 1]], dtype=float))
    Ap = A.copy(order="F")
    sd.swap22(Ap, 0)
    assert (A == Ap).all()
    sd.swap12(Ap, 0)
    assert (A == Ap).all()
    sd.swap21(Ap, 0)
    assert (A == Ap).all()


# ------------------------------------------------------------------ test/collect_eigen.jl
@pytest.mark.parametrize("dtype", DTYPES)
def test_collect_eigen_triangular(dtype):
    rng = np.random.default_rng(16)
    n = 20
    R = np.asfortranarray(np.triu(rnd(rng, dtype, n, n)))
    lam, xs = sla.eig(R)
    assert np.allclose(lam, np.diag(R))  # LAPACK keeps the diagonal order for triangular input
    x = np.zeros(n, dtype=np.complex128)
    for i in range(n):
        x[:] = 0
        sd.collect_eigen(x, R, i)
        assert np.linalg.norm(x) == pytest.approx(1.0)
        np.testing.assert_allclose(np.abs(x), np.abs(xs[:, i]), atol=1e-9)


def rot(t):
    return np.array([[np.cos(t), np.sin(t)], [-np.sin(t), np.cos(t)]])


def test_collect_eigen_quasi_triangular():
    rng = np.random.default_rng(17)
    n = 20
    R = np.asfortranarray(np.triu(rng.random((n, n))))
    R[0:2, 0:2] = rot(1.0) + np.eye(2)
    R[9:11, 9:11] = rot(6 / 5) + 2 * np.eye(2)
    lam, xs = sla.eig(R)
    x = np.zeros(n, dtype=np.complex128)
    mine = sd.eigenvalues(R)
    for i in range(n):
        x[:] = 0
        ln = sd.collect_eigen(x, R, i)
        assert np.linalg.norm(x) == pytest.approx(1.0)
        # residual of the eigenpair instead of relying on LAPACK's column order
        lam_i = mine[i] if not (i in (1, 10)) else mine[i - 1]
        assert np.linalg.norm(R @ x - lam_i * x) < 1e-10
        j = int(np.argmin(np.abs(lam - lam_i)))
        np.testing.assert_allclose(np.abs(x), np.abs(xs[:, j]), atol=1e-8)


def test_copy_eigenvalues_partial():  # test/collect_eigen.jl:67-78
    rng = np.random.default_rng(18)
    R = np.asfortranarray(np.triu(rng.random((20, 20))))
    R[0:2, 0:2] = rot(1.0) + np.eye(2)
    for last in (2, 3):
        lam = np.linalg.eigvals(R[: last + 1, : last + 1])
        th = sd.copy_eigenvalues(np.zeros(last + 1, dtype=complex), R, 0, last)
        np.testing.assert_allclose(reim_sorted(lam), reim_sorted(th), atol=1e-13)


# ------------------------------------------------------------------ test/householder.jl
@pytest.mark.parametrize("dtype", DTYPES)
def test_reflector(dtype):
    rng = np.random.default_rng(19)
    n = 20
    x = rnd(rng, dtype, n)
    z = x.copy()
    tau = sd.reflector(z, n)
    z[n - 1] = 1
    y = x - tau * np.vdot(z, x) * z
    assert np.linalg.norm(y[: n - 1]) <= 10 * EPS
    assert abs(y[n - 1].real) == pytest.approx(np.linalg.norm(x))
    assert abs(np.imag(y[n - 1])) <= 4 * EPS
    assert 1 <= np.real(tau) <= 2 and abs(tau - 1) <= 1 + 1e-15
    assert sd.reflector(np.array([0, 0, 5], dtype=dtype), 3) == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_reflector_lmul_rmul(dtype):
    rng = np.random.default_rng(20)
    A = rnd(rng, dtype, 4, 4)
    G = sd.Reflector(4, dtype)
    G.vec[:] = rnd(rng, dtype, 4)
    tau = G.build(4)
    z = np.concatenate([G.vec[:3], [1]])
    Hm = np.eye(4) - tau * np.outer(z, z.conj())
    B = A.copy()
    sd.reflector_rmul(B, G, 0, 3)
    np.testing.assert_allclose(B, A @ Hm.conj().T, atol=1e-14)
    B = A.copy()
    sd.reflector_lmul(G, B, 0, 3)
    np.testing.assert_allclose(B, Hm @ A, atol=1e-14)


@pytest.mark.parametrize("dtype", DTYPES)
def test_restore_arnoldi(dtype):
    """Spec of test/householder.jl:68-88 (`A*W[:,1:k] = W*H` to 1e-14 after restore_arnoldi!).

    That (no longer run) reference test feeds a FULL `Q'HQ`; the current
    `restore_arnoldi!` (src/restore_hessenberg.jl:89-96) restricts its Givens sweep to
    rows 1..min(i+2,to), i.e. it assumes the (quasi-)triangular Schur form the driver
    always hands it (src/run.jl:281,355,360).  So the change of basis here is the Schur
    factorisation itself."""
    from oracle import arnoldi as oa

    rng = np.random.default_rng(21)
    n, k = 10, 6
    A = rnd(rng, dtype, n, n)
    ws = oa.ArnoldiWorkspace.from_dims(dtype, n, k)
    oa.reinitialize(ws, 0)
    oa.iterate_arnoldi(A, ws, 1, k)
    H = ws.H.copy(order="F")
    Q = np.eye(k, dtype=dtype, order="F")
    assert sd.local_schurfact(H[:k, :], 0, k - 1, Q)
    sd.restore_arnoldi(H, 0, k - 1, Q, sd.Reflector(k, dtype))
    W = np.hstack([ws.V[:, :k] @ Q, ws.V[:, k : k + 1]])
    assert np.linalg.norm(A @ W[:, :k] - W @ H) < 1e-13
    assert np.linalg.norm(np.tril(H[:k, :k], -2)) < 1e-14
    assert np.linalg.norm(Q.conj().T @ Q - np.eye(k)) < 100 * EPS


@pytest.mark.parametrize("dtype", DTYPES)
def test_restore_arnoldi_partial_range(dtype):
    """Same invariant when only columns from..to (a locked prefix before, purged tail after)
    are restored, as src/run.jl:360 does with from = nlock+1, to = k."""
    from oracle import arnoldi as oa

    rng = np.random.default_rng(22)
    n, m = 12, 8
    A = rnd(rng, dtype, n, n)
    ws = oa.ArnoldiWorkspace.from_dims(dtype, n, m)
    oa.reinitialize(ws, 0)
    oa.iterate_arnoldi(A, ws, 1, m)
    H = ws.H.copy(order="F")
    Q = np.eye(m, dtype=dtype, order="F")
    assert sd.local_schurfact(H[:m, :], 0, m - 1, Q)
    # keep the first `to+1` Schur vectors; make sure we do not cut a 2x2 block
    to = 5 if H[6, 5] == 0 else 6
    sd.restore_arnoldi(H, 0, to, Q, sd.Reflector(m, dtype))
    k = to + 1
    W = np.hstack([ws.V[:, :m] @ Q[:, :k], ws.V[:, m : m + 1]])
    assert np.linalg.norm(A @ W[:, :k] - W @ H[: k + 1, :k]) < 1e-12
