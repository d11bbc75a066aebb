"""`-m gpu`: the device layouts of a stored matrix (mul!(y, A, x), src/expansion.jl:121) -- CSR row blocks
(skewed rows, rows longer than a block, 64-bit offsets), sliced ELLPACK (padding, sigma-window sorting), the
delta-value-indexed kernel's rows-per-thread variants -- against scipy and against each other BIT for bit
(every layout rounds each product on its own and adds in CSR order).  All through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from __graft_entry__ import import_package
from oracle import arnoldi as oa
from oracle.matrices import laplace3d

pytestmark = pytest.mark.gpu
pkg = import_package()
DTYPES = [np.float64, np.complex128]


def rnd(rng, dtype, *shape):
    a = rng.standard_normal(shape)
    if np.dtype(dtype).kind == "c":
        a = a + 1j * rng.standard_normal(shape)
    return a.astype(dtype)


def _skewed(rng, dtype, n):
    """Short random rows + empty rows + a band of 300-entry rows + rows far longer than any block capacity."""
    cplx = np.dtype(dtype).kind == "c"
    A = sp.random(n, n, density=4.0 / n, random_state=rng, format="lil", dtype=np.float64)
    for r, cnt in ((3, 4097), (n // 3, 9000), (n - 2, 9000)):
        c = rng.choice(n, cnt, replace=False)
        A[r, c] = rng.standard_normal(cnt)
    for r in range(n // 2, n // 2 + 40):
        c = rng.choice(n, 300, replace=False)
        A[r, c] = rng.standard_normal(300)
    A[10:30, :] = 0
    A = A.tocsr()
    if cplx:
        B = A.copy()
        B.data = rng.standard_normal(B.nnz)
        A = (A + 1j * B).tocsr()
    A.sort_indices()
    return A.astype(dtype)


def _apply(A, x, dtype, ctx=None):
    op = pkg.csr_operator(A, ctx)
    ws = pkg.ArnoldiWorkspace(A.shape[0], 2, dtype, ctx=op.ctx)
    ws.set_col(0, x)
    ws.apply(op, 0, 1)
    return ws.col(1), op.format, op


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ptr64", ["0", "1"])
def test_csr_row_blocks_long_rows_and_64bit_offsets(dtype, ptr64, monkeypatch):
    """k_spmv_csr on a skewed matrix: blocks closed by the non-zero budget, rows with more entries than a block holds
    (all-threads path), empty rows -- vs scipy to 1e-13 relative; the int64-offset instantiation (what a matrix with
    nnz >= 2^31 gets, forced by KS_SPMV_PTR64=1) must be BIT-identical to the int32 one; and the same matrix handed
    over as CSC / Int64 / 1-based (Julia's SparseMatrixCSC) builds the same operator."""
    rng = np.random.default_rng(101)
    n = 20011
    A = _skewed(rng, dtype, n)
    x = rnd(rng, dtype, n)
    monkeypatch.setenv("KS_SPMV_PTR64", ptr64)
    monkeypatch.setenv("KS_SPMV_FORMAT", "csr")
    y, fmt, op = _apply(A, x, dtype)
    assert fmt["layout"] == "csr"
    ref = A @ x
    scale = np.abs(A) @ np.abs(x)
    assert np.all(np.abs(y - ref) <= 1e-13 * (scale + 1e-300))
    assert np.all(y[10:30] == 0)
    monkeypatch.setenv("KS_SPMV_PTR64", "0")
    y0, _, _ = _apply(A, x, dtype, op.ctx)
    assert np.array_equal(y0, y)
    Ac = A.tocsc()
    L = pkg._lib.load()
    h = C.c_void_p()
    ptr, idx, val = (Ac.indptr.astype(np.int64) + 1), (Ac.indices.astype(np.int64) + 1), np.ascontiguousarray(Ac.data)
    pkg._lib.check(L.ks_operator_csr(op.ctx._h, n, n, Ac.nnz, ptr.ctypes.data, idx.ctypes.data, val.ctypes.data, pkg._lib.KS_CSC, 1,
                                     pkg._lib.KS_I64, pkg._lib.KS_C64 if np.dtype(dtype).kind == "c" else pkg._lib.KS_F64, C.byref(h)))
    opj = pkg.Operator(op.ctx, h, (n, n), dtype)
    ws = pkg.ArnoldiWorkspace(n, 2, dtype, ctx=op.ctx)
    ws.set_col(0, x)
    ws.apply(opj, 0, 1)
    assert np.array_equal(ws.col(1), y)


@pytest.mark.parametrize("dtype", DTYPES)
def test_sliced_ellpack_is_bit_identical_and_chosen_for_uniform_rows(dtype, monkeypatch):
    """A 7-point stencil with VARIABLE coefficients (no dictionary layout applies) is stored as sliced ELLPACK by
    default; y must equal the CSR-block kernel's bit for bit.  Forced on a ragged matrix (heavy padding, empty rows,
    Inf in x next to padding) with and without sigma-window sorting it must still agree; a value-indexed SELL too."""
    cplx = np.dtype(dtype).kind == "c"
    rng = np.random.default_rng(55)
    A = laplace3d(13, 14, 15).astype(dtype)
    A.data = A.data * (1.0 + 0.5 * rng.random(A.nnz)) + (0.1j * rng.random(A.nnz) if cplx else 0)
    n = A.shape[0]
    x = rnd(rng, dtype, n)
    y, fmt, op = _apply(A, x, dtype)
    assert fmt["layout"] == "sell" and fmt["ndict"] == 0 and fmt["bytes_per_nnz"] < 1.15 * (4 + np.dtype(dtype).itemsize)
    monkeypatch.setenv("KS_SPMV_FORMAT", "csr")
    y_csr, f_csr, _ = _apply(A, x, dtype, op.ctx)
    assert f_csr["layout"] == "csr" and np.array_equal(y.view(np.uint64), y_csr.view(np.uint64))
    np.testing.assert_allclose(y, A @ x, rtol=1e-13, atol=1e-13)
    # ragged: rows of 0..40 entries, a few distinct values
    m = 5003
    R = sp.random(m, m, density=8.0 / m, random_state=rng, format="csr", dtype=np.float64)
    R.data = np.array([1.5, -2.0, 0.25, -0.0])[rng.integers(0, 4, R.nnz)]
    R = R.tolil()
    R[100:164, :] = 0
    R[7, rng.choice(m, 40, replace=False)] = 3.0
    R = R.tocsr().astype(dtype)
    R.sort_indices()
    xr = rnd(rng, dtype, m)
    xr[rng.choice(m, 5, replace=False)] = np.inf  # a padding entry must never be multiplied
    ref, _, op2 = _apply(R, xr, dtype)
    assert _["layout"] == "csr"
    for fmt_name, sigma in (("sell", "1"), ("sell", "256"), ("sellvi", "1"), ("sellvi", "640")):
        monkeypatch.setenv("KS_SPMV_FORMAT", fmt_name)
        monkeypatch.setenv("KS_SELL_SIGMA", sigma)
        got, f, _o = _apply(R, xr, dtype, op2.ctx)
        assert f["layout"] == ("sell-vi" if fmt_name == "sellvi" else "sell"), f
        assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), (fmt_name, sigma)
    monkeypatch.setenv("KS_SPMV_PTR64", "1")
    got, f, _o = _apply(R, xr, dtype, op2.ctx)
    assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))


@pytest.mark.parametrize("rpt", ["1", "2", "4"])
@pytest.mark.parametrize("shape", [(37, 41, 43), (5, 3, 2), (300, 7, 1)])
def test_dvi_rows_per_thread_variants_bit_identical(rpt, shape, monkeypatch):
    """k_spmv_dvi with 1 / 2 / 4 rows per thread (LDS-staged codes) vs the CSR blocks and sliced ELLPACK: bit-identical
    y, including grids whose row count is not a multiple of the tile and tiles with ragged rows (stencil boundaries)."""
    A = laplace3d(*shape)
    n = A.shape[0]
    x = oa.uniform_hash(9, np.arange(n)) - 0.5
    monkeypatch.setenv("KS_SPMV_FORMAT", "csr")
    y0, f0, op0 = _apply(A, x, np.float64)
    assert f0["bytes_per_nnz"] == 12.0
    monkeypatch.setenv("KS_SPMV_FORMAT", "dvi")
    monkeypatch.setenv("KS_DVI_RPT", rpt)
    y1, f1, _ = _apply(A, x, np.float64, op0.ctx)
    assert f1["layout"] == "csr-dvi" and f1["bytes_per_nnz"] == 1.0
    assert np.array_equal(y1, y0)
    monkeypatch.setenv("KS_SPMV_FORMAT", "sellvi")
    y2, f2, _ = _apply(A, x, np.float64, op0.ctx)
    assert f2["layout"] == "sell-vi" and np.array_equal(y2, y0)
    monkeypatch.setenv("KS_SPMV_FORMAT", "stencil")
    y3, f3, _ = _apply(A, x, np.float64, op0.ctx)
    assert f3["layout"] == "stencil" and f3["ndict"] <= 7 and np.array_equal(y3, y0)
    np.testing.assert_allclose(y0, A @ x, rtol=0, atol=1e-14 * 12)


def test_solver_end_to_end_on_each_layout(monkeypatch):
    """The layout must not change what the solver does: same mat-vec count and Ritz values on every layout."""
    A = laplace3d(14, 15, 16)
    n = A.shape[0]
    v1 = oa.uniform_hash(oa.DEFAULT_SEED, np.arange(n))
    out = {}
    for f in ("stencil", "dvi", "vi", "csr", "sell", "sellvi"):
        monkeypatch.setenv("KS_SPMV_FORMAT", f)
        dec, hist = pkg.partialschur(A, v1=v1, nev=5, which="SR", tol=1e-10, maxdim=25)
        out[f] = (hist.mvproducts, np.sort(dec.eigenvalues.real))
        assert hist.converged
    for f in out:
        assert out[f][0] == out["csr"][0] and np.array_equal(out[f][1], out["csr"][1]), f


@pytest.mark.parametrize("dtype", DTYPES)
def test_stencil_mask_layout(dtype, monkeypatch):
    """One bit per dictionary slot and row.  (i) 27-point-like stencil with 19 slots (32-bit masks), complex values,
    boundary rows = sub-sequences: bit-identical to the CSR blocks.  (ii) x containing Inf next to ABSENT slots: a
    skipped slot must never be multiplied.  (iii) a matrix whose rows cannot be embedded in one entry order (two rows
    using the same two dictionary entries in opposite orders) falls back to the delta-value-indexed layout;
    KS_SPMV_FORMAT=stencil then refuses.  (iv) 33 dictionary entries: too many slots -> DVI."""
    cplx = np.dtype(dtype).kind == "c"
    rng = np.random.default_rng(91)
    mx, my, mz = 9, 8, 7
    n = mx * my * mz
    rows, cols, vals = [], [], []
    offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if abs(dx) + abs(dy) + abs(dz) <= 2]
    coef = {o: (rng.standard_normal() + (1j * rng.standard_normal() if cplx else 0)) for o in offs}
    for z in range(mz):
        for y in range(my):
            for x_ in range(mx):
                r = x_ + mx * (y + my * z)
                for (dx, dy, dz) in offs:
                    if 0 <= x_ + dx < mx and 0 <= y + dy < my and 0 <= z + dz < mz:
                        rows.append(r); cols.append(r + dx + mx * (dy + my * dz)); vals.append(coef[(dx, dy, dz)])
    A = sp.csr_matrix((np.array(vals, dtype=dtype), (rows, cols)), shape=(n, n))
    A.sort_indices()
    xv = rnd(rng, dtype, n)
    y, f, op = _apply(A, xv, dtype)
    assert f["layout"] == "stencil" and f["ndict"] == len(offs) == 19
    monkeypatch.setenv("KS_SPMV_FORMAT", "csr")
    y0, f0, _ = _apply(A, xv, dtype, op.ctx)
    assert f0["layout"] == "csr" and np.array_equal(y.view(np.uint64), y0.view(np.uint64))
    # (ii)
    monkeypatch.delenv("KS_SPMV_FORMAT")
    L = laplace3d(6, 5, 4).astype(dtype)
    m = L.shape[0]
    xi = rnd(rng, dtype, m)
    xi[5] = np.inf                      # x index 5 = right neighbour of row 4... and NOT a neighbour of row 6 (x = 0 of the next line)
    yi, fi, opi = _apply(L, xi, dtype)
    assert fi["layout"] == "stencil"
    ref = L @ xi
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(yi), fin) and np.allclose(yi[fin], ref[fin], rtol=1e-13, atol=1e-13)
    # (iii)
    B = sp.csr_matrix(np.array([[0, 2.0, 0, 3.0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0.0]]))  # row 0: deltas (+1: 2.0), (+3: 3.0)
    B = sp.lil_matrix((6, 6))
    B[0, 1], B[0, 3] = 2.0, 3.0        # entries (delta 1, 2.0) then (delta 3, 3.0)
    B[2, 3], B[2, 5] = 3.0, 2.0        # entries (delta 1, 3.0) then (delta 3, 2.0): fine so far (4 distinct entries)
    B[3, 4], B[3, 5] = 2.0, 9.0        # (delta 1, 2.0) then (delta 2, 9.0)
    B[1, 3], B[1, 4] = 9.0, 3.0        # (delta 2, 9.0) then (delta 3, 3.0)  -> 2.0@1 < 9.0@2 < 3.0@3 consistent
    C2 = sp.lil_matrix((6, 6))
    C2[0, 1], C2[0, 2] = 5.0, 7.0      # (delta 1, 5.0) precedes (delta 2, 7.0)
    C2[3, 5] = 7.0                      # (delta 2, 7.0) alone
    C2[2, 4], C2[2, 3] = 7.0, 5.0      # row 2: (delta 1, 5.0) then (delta 2, 7.0) again -- consistent; make a cycle instead:
    D = sp.csr_matrix((np.array([5.0, 7.0, 7.0, 5.0]), (np.array([0, 0, 2, 2]), np.array([1, 2, 3, 5]))), shape=(6, 6))
    # row 0: (delta 1, 5.0), (delta 2, 7.0);  row 2: (delta 1, 7.0), (delta 3, 5.0) -> four distinct entries, still acyclic
    E = sp.csr_matrix((np.array([5.0, 7.0, 7.0, 5.0]), (np.array([0, 0, 3, 3]), np.array([1, 2, 4, 5]))), shape=(6, 6))
    # row 0: (d1, 5), (d2, 7);  row 3: (d1, 7), (d2, 5): distinct entries (d1,5) (d2,7) (d1,7) (d2,5): acyclic as well.
    # A genuine conflict needs the SAME two entries in both orders, which sorted columns forbid (same deltas, same order);
    # unsorted CSR input can do it:
    ptr = np.array([0, 2, 2, 4, 4, 4, 4], dtype=np.int64)
    idx = np.array([1, 2, 4, 3], dtype=np.int64)            # row 0: cols 1, 2;  row 2: cols 4, 3 (unsorted)
    val = np.array([5.0, 7.0, 7.0, 5.0]).astype(dtype)       # row 0: (d1,5),(d2,7); row 2: (d2,7),(d1,5)  -> cycle
    import ctypes as C

    Lh = pkg._lib.load()
    h = C.c_void_p()
    code = pkg._lib.KS_C64 if cplx else pkg._lib.KS_F64
    pkg._lib.check(Lh.ks_operator_csr(op.ctx._h, 6, 6, 4, ptr.ctypes.data, idx.ctypes.data, val.ctypes.data, pkg._lib.KS_CSR, 0, pkg._lib.KS_I64, code, C.byref(h)))
    opc = pkg.Operator(op.ctx, h, (6, 6), dtype)
    assert opc.format["layout"] == "csr-dvi"
    ws6 = pkg.ArnoldiWorkspace(6, 2, dtype, ctx=op.ctx)
    x6 = rnd(rng, dtype, 6)
    ws6.set_col(0, x6)
    ws6.apply(opc, 0, 1)
    want = np.zeros(6, dtype=dtype)
    want[0] = 5.0 * x6[1] + 7.0 * x6[2]
    want[2] = 7.0 * x6[4] + 5.0 * x6[3]
    np.testing.assert_allclose(ws6.col(1), want, rtol=1e-14)
    monkeypatch.setenv("KS_SPMV_FORMAT", "stencil")
    h2 = C.c_void_p()
    rc = Lh.ks_operator_csr(op.ctx._h, 6, 6, 4, ptr.ctypes.data, idx.ctypes.data, val.ctypes.data, pkg._lib.KS_CSR, 0, pkg._lib.KS_I64, code, C.byref(h2))
    assert rc == pkg._lib.KS_ERR_ARGUMENT
    monkeypatch.delenv("KS_SPMV_FORMAT")
    # (iv) a banded matrix with 33 distinct diagonals
    n4 = 400
    diags = [np.full(n4 - k, 1.0 + k) for k in range(33)]
    G = sp.diags(diags, list(range(33)), format="csr").astype(dtype)
    y4, f4, _ = _apply(G, rnd(rng, dtype, n4), dtype, op.ctx)
    assert f4["layout"] == "csr-dvi" and f4["ndict"] == 33


def test_malformed_pointer_arrays_are_rejected():
    """a non-monotone or out-of-range pointer array must be refused at upload (it would index host arrays
    during the CSC conversion, or device arrays in the SpMV, out of bounds)."""
    L = pkg._lib.load()
    ctx = pkg.default_context()
    val = np.ones(4)
    idx = np.array([0, 1, 2, 3], dtype=np.int64)
    for layout in (pkg._lib.KS_CSR, pkg._lib.KS_CSC):
        for ptr in ([0, 3, 2, 4, 4], [0, 1, 5, 3, 4], [1, 1, 2, 3, 4], [0, 1, 2, 3, 3]):
            p = np.array(ptr, dtype=np.int64)
            h = C.c_void_p()
            rc = L.ks_operator_csr(ctx._h, 4, 4, 4, p.ctypes.data, idx.ctypes.data, val.ctypes.data, layout, 0, pkg._lib.KS_I64, pkg._lib.KS_F64, C.byref(h))
            assert rc == pkg._lib.KS_ERR_ARGUMENT, (layout, ptr)
    bad_col = np.array([0, 1, 2, 7], dtype=np.int64)
    p = np.array([0, 1, 2, 3, 4], dtype=np.int64)
    h = C.c_void_p()
    assert L.ks_operator_csr(ctx._h, 4, 4, 4, p.ctypes.data, bad_col.ctypes.data, val.ctypes.data, pkg._lib.KS_CSR, 0, pkg._lib.KS_I64, pkg._lib.KS_F64, C.byref(h)) == pkg._lib.KS_ERR_ARGUMENT


@pytest.mark.parametrize("dtype", DTYPES)
def test_column_blocked_csr_is_bit_identical(dtype, monkeypatch):
    """KS_LAYOUT_CSR_CB: the matrix split into column blocks, one launch each, the row sums continued from launch to launch
    in CSR order -- y must equal the plain CSR row-block layout BIT for bit, for 2, 3 and 5 blocks, with empty rows, rows
    confined to one block and rows that span all of them; the auto rule picks it for config 3's matrix (n = 1e6: x = 8 MB,
    scattered columns) and leaves a banded matrix alone."""
    rng = np.random.default_rng(11)
    n = 30_000
    A = sp.random(n, n, density=6.0 / n, random_state=rng, format="lil", dtype=np.float64)
    A[100:140, :] = 0                                   # empty rows
    for r in range(200, 260):                           # rows confined to the first / last column block
        A[r, :] = 0
        A[r, rng.choice(n // 8, 5, replace=False)] = rng.standard_normal(5)
        A[r + 100, :] = 0
        A[r + 100, n - 1 - rng.choice(n // 8, 5, replace=False)] = rng.standard_normal(5)
    A = A.tocsr()
    if np.dtype(dtype).kind == "c":
        B = A.copy()
        B.data = rng.standard_normal(B.nnz)
        A = (A + 1j * B).tocsr()
    A = A.astype(dtype)
    A.sort_indices()
    x = rnd(rng, dtype, n)
    monkeypatch.setenv("KS_SPMV_FORMAT", "csr")
    monkeypatch.setenv("KS_SPMV_COLBLOCKS", "0")
    y0, f0, _ = _apply(A, x, dtype)
    assert f0["layout"] == "csr"
    for nb in ("2", "3", "5"):
        monkeypatch.setenv("KS_SPMV_COLBLOCKS", nb)
        y, f, _ = _apply(A, x, dtype)
        assert f["layout"] == "csr-cb", f
        assert np.array_equal(y, y0), nb
    # the SINGLE-LAUNCH form (k_spmv_csr_cb: a workgroup keeps its rows' sums in registers while it walks the column blocks;
    # automatic from 5 blocks on): every rows-per-workgroup variant, 2 / 5 / 8 blocks, against one launch per block
    for nb in ("2", "5", "8"):
        monkeypatch.setenv("KS_SPMV_COLBLOCKS", nb)
        for rpt in ("1", "2", "4", "8", "16"):
            monkeypatch.setenv("KS_SPMV_CB_RPT", rpt)
            y, f, _ = _apply(A, x, dtype)
            assert f["layout"] == "csr-cb" and np.array_equal(y, y0), (nb, rpt)
        monkeypatch.delenv("KS_SPMV_CB_RPT")
        monkeypatch.setenv("KS_SPMV_CB_SINGLE", "0")
        y, f, _ = _apply(A, x, dtype)
        assert np.array_equal(y, y0), nb
        monkeypatch.delenv("KS_SPMV_CB_SINGLE")
    # unsorted rows: the order of the additions would change -> the layout must refuse
    monkeypatch.setenv("KS_SPMV_COLBLOCKS", "2")
    U = A.copy()
    U.has_sorted_indices = False
    r = 5000
    a, b = U.indptr[r], U.indptr[r + 1]
    if b - a >= 2:
        U.indices[a:b] = U.indices[a:b][::-1].copy()
        U.data[a:b] = U.data[a:b][::-1].copy()
        yu, fu, _ = _apply(U, x, dtype)
        assert fu["layout"] == "csr"
    monkeypatch.delenv("KS_SPMV_FORMAT")
    monkeypatch.delenv("KS_SPMV_COLBLOCKS")
    if np.dtype(dtype).kind == "f":
        H = pkg.matrices.hashed_nonsymmetric_csr(1_000_000, seed=7)
        xh = rnd(rng, dtype, H.shape[0])
        yh, fh, _ = _apply(H, xh, dtype)
        assert fh["layout"] == "csr-cb", fh
        monkeypatch.setenv("KS_SPMV_COLBLOCKS", "0")
        yp, fp, _ = _apply(H, xh, dtype)
        assert fp["layout"] == "csr" and np.array_equal(yh, yp)
        monkeypatch.delenv("KS_SPMV_COLBLOCKS")
        Bd = sp.diags([rng.standard_normal(1_000_000 - abs(k)) for k in (-3, -1, 0, 1, 3)], [-3, -1, 0, 1, 3], format="csr")
        _, fb, _ = _apply(Bd, xh, dtype)
        assert fb["layout"] != "csr-cb", fb


@pytest.mark.parametrize("grid", [(182, 182, 9), (181, 182, 10), (256, 128, 8), (64, 64, 70)])
def test_marching_forms_of_the_stencil_product_are_bit_identical(grid, monkeypatch):
    """Float64 stencil-mask products on one GPU go through the persistent kernels of csrc/ks_spmv_march.hpp: the z-marching form
    where a plane has >= 64 tiles of 512 rows and there are >= 8 planes (the first three grids: nx even, nx odd -- the near taps
    +-nx then sit at odd offsets --, a power-of-two plane), the window form for smaller planes (the last grid), the register form
    behind them.  Every form, k_spmv_stencil2 and the CSR row blocks must give the SAME bits (products rounded separately, added in
    slot order under the row's mask), for the plain product and -- through two block cycles of the expansion, whose Newton steps
    y = sigma (A x - theta x) are fused into these kernels -- for the shifted one: identical Hessenberg matrices, bit for bit."""
    from test_gpu_sstep import _lockstep

    mx, my, mz = grid
    A = laplace3d(mx, my, mz)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    xv = rnd(rng, np.float64, n)
    forms = {"default": {}, "window": {"KS_MARCH_Z": "0"}, "registers": {"KS_MARCH_Z": "0", "KS_MARCH_WINDOW": "0"}, "stencil2": {"KS_STENCIL_MARCH": "0"},
             "z-ranges 5": {"KS_MARCH_ZR": "5"}}
    ys, Hs = {}, {}
    ctx = None
    for name, env in forms.items():
        for k_ in ("KS_MARCH_Z", "KS_MARCH_WINDOW", "KS_STENCIL_MARCH", "KS_MARCH_ZR"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        y, f, op = _apply(A, xv, np.float64, ctx)
        ctx = op.ctx
        assert f["layout"] == "stencil"
        ys[name] = y
        for cyc, _Hs, Hb, _Vs, _Vb, rel, orth, info in _lockstep(A, np.float64, 8, 6, 8, 16, "SR", 2):
            if cyc == 1:
                assert info["blocks"] > 0 and info["abandoned"] == 0, (name, info)
                Hs[name] = Hb
    ref = A @ xv
    np.testing.assert_allclose(ys["default"], ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    for name in forms:
        assert np.array_equal(ys[name].view(np.uint64), ys["default"].view(np.uint64)), name
        assert np.array_equal(Hs[name].view(np.uint64), Hs["default"].view(np.uint64)), name
    monkeypatch.delenv("KS_MARCH_ZR", raising=False)
    monkeypatch.setenv("KS_SPMV_FORMAT", "csr")
    y0, f0, _ = _apply(A, xv, np.float64, ctx)
    assert f0["layout"] == "csr" and np.array_equal(y0.view(np.uint64), ys["default"].view(np.uint64))
