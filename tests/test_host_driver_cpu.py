"""`-m "not gpu"`: the product's C++ driver (`_partialschur` restated in ks_driver.hpp) run end to end
on the ORACLE's OpenMP CPU backend (oracle/cpu_backend.cpp -- test infrastructure, never part of the
product library) and compared with the Python oracle on identical inputs: same mat-vec counts on the
deterministic KATs, same eigenvalues to rounding."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import arnoldi as oa
from oracle import cpuref
from oracle.matrices import hashed_nonsymmetric, laplace1d, laplace3d, laplace3d_eigs

EPS = np.finfo(np.float64).eps


def both(A, **kw):
    n = A.shape[0]
    v1 = oa.uniform_hash(oa.DEFAULT_SEED, np.arange(n))
    if A.dtype.kind == "c":
        v1 = v1 + 0j
    r = cpuref.partialschur_csr(A, v1=v1, **kw)
    dec, hist = oa.partialschur(A, v1=v1, **kw)
    return r, dec, hist


def test_readme_example_same_counts():
    A = laplace1d(100)
    r, dec, hist = both(A, nev=10, which="SR", tol=1e-6)
    assert r["converged"] and r["mvproducts"] == hist.mvproducts and r["restarts"] == hist.restarts
    np.testing.assert_allclose(r["eigenvalues"], dec.eigenvalues, atol=1e-12)
    assert np.linalg.norm(A @ r["Q"] - r["Q"] @ r["R"]) < 1e-6


def test_laplace3d_same_counts():
    A = laplace3d(8, 9, 10)
    r, dec, hist = both(A, nev=6, which="SR", tol=1e-10, maxdim=30)
    assert r["mvproducts"] == hist.mvproducts
    np.testing.assert_allclose(np.sort(r["eigenvalues"].real), laplace3d_eigs(8, 9, 10)[:6], atol=1e-9)
    assert np.linalg.norm(r["Q"].T @ r["Q"] - np.eye(6)) < 100 * EPS


def test_nonsymmetric_pairs():
    planted = [(5.0, 3.0), (4.0, -2.5), (-6.0, 1.0), (7.5, 0.0)]
    A = hashed_nonsymmetric(300, seed=7, planted=planted)
    r, dec, hist = both(A, nev=6, which="LM", tol=1e-10)
    assert r["converged"] and r["nconverged"] == hist.nconverged and r["mvproducts"] == hist.mvproducts
    np.testing.assert_allclose(np.sort_complex(r["eigenvalues"]), np.sort_complex(dec.eigenvalues), atol=1e-9)
    assert np.linalg.norm(A @ r["Q"] - r["Q"] @ r["R"]) < 1e-8


def test_complex():
    d = np.arange(1, 81) * (1 + 0.25j)
    A = (sp.diags(d) + 0.01 * (hashed_nonsymmetric(80, seed=11) + 1j * hashed_nonsymmetric(80, seed=12))).tocsr()
    r, dec, hist = both(A, nev=5, which="LR", tol=1e-10)
    assert r["converged"] and r["mvproducts"] == hist.mvproducts
    np.testing.assert_allclose(r["eigenvalues"], dec.eigenvalues, atol=1e-9)
    assert np.linalg.norm(A @ r["Q"] - r["Q"] @ r["R"]) < 1e-8


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_kat2_counts(dtype):
    """Deterministic mat-vec counts of test/partial_schur.jl: 7 (rank 3), 5 (zero matrix)."""
    rng = np.random.default_rng(7)
    X = rng.random((10, 3)) + (1j * rng.random((10, 3)) if np.dtype(dtype).kind == "c" else 0)
    B = sp.csr_matrix(X @ X.conj().T)
    r = cpuref.partialschur_csr(B, nev=5, mindim=5, maxdim=7, tol=EPS)
    assert r["converged"] and r["mvproducts"] == 7
    Z = sp.csr_matrix((5, 5), dtype=dtype)
    r = cpuref.partialschur_csr(Z, nev=5)
    assert r["converged"] and r["mvproducts"] == 5 and r["nconverged"] == 5
    assert np.linalg.norm(r["Q"].conj().T @ r["Q"] - np.eye(5)) < 100 * EPS


def test_argument_errors_from_cxx():
    A = laplace1d(6)
    with pytest.raises(ValueError):
        cpuref.partialschur_csr(A, nev=5, mindim=3, maxdim=6)
    with pytest.raises(ValueError):
        cpuref.partialschur_csr(A, nev=0)


def test_timed_cycles_sample_runs():
    A = laplace3d(12, 12, 12)
    t = cpuref.timed_cycles_csr(A, nev=20, which="SR", mindim=20, maxdim=40, cycles=2)
    assert t["steps"] > 0 and t["seconds"] > 0 and t["threads"] >= 1
