"""`-m "not gpu"`: the product's numpy CSR generators vs the oracle's independent scipy constructions,
and the shared counter-based RNG."""
import numpy as np

from __graft_entry__ import import_package
from oracle import arnoldi as oa
from oracle import matrices as om

pkg = import_package()
pm = pkg.matrices


def test_rng_identical():
    idx = np.arange(5000)
    assert (pm.uniform_hash(20240917, idx) == oa.uniform_hash(20240917, idx)).all()
    assert (pm.start_vector(100, row_begin=40) == oa.uniform_hash(20240917, np.arange(40, 140))).all()


def test_laplace3d_matches_kron_construction():
    for dims in [(5, 6, 7), (1, 4, 3), (8, 1, 1), (4, 4, 4)]:
        ip, ix, dv = pm.laplace3d_csr(*dims)
        A = pm.to_scipy(ip, ix, dv, np.prod(dims))
        B = om.laplace3d(*dims)
        assert abs(A - B).nnz == 0
        assert ix.dtype == np.int32
    # row slabs (what each rank of a distributed run generates) stack to the whole matrix
    n = 5 * 6 * 7
    parts = [pm.to_scipy(*pm.laplace3d_csr(5, 6, 7, r0, r1), n) for r0, r1 in [(0, 60), (60, 150), (150, n)]]
    import scipy.sparse as sp

    assert abs(sp.vstack(parts) - om.laplace3d(5, 6, 7)).nnz == 0


def test_laplace1d_and_eigs():
    ip, ix, dv = pm.laplace1d_csr(50)
    assert abs(pm.to_scipy(ip, ix, dv, 50) - om.laplace1d(50)).nnz == 0
    np.testing.assert_allclose(pm.laplace3d_eigs(4, 5, 6), om.laplace3d_eigs(4, 5, 6))
    np.testing.assert_allclose(np.linalg.eigvalsh(om.laplace3d(3, 4, 5).toarray()), om.laplace3d_eigs(3, 4, 5), atol=1e-12)


def test_hashed_nonsymmetric_identical():
    A = pm.hashed_nonsymmetric_csr(500, seed=7)
    B = om.hashed_nonsymmetric(500, seed=7)
    assert abs(A - B).nnz == 0
    deg = np.diff(A.indptr)
    assert 1 <= deg.min() and deg.max() <= 9 and 4 < deg.mean() < 6
