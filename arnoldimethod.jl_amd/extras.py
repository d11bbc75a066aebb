"""Ready-made DEVICE operators built on the opaque-operator seam (`ks_operator_device_callback`, include/kschur.h;
`mul!(y, A, x)` with a user-supplied `A`, src/expansion.jl:121).  These are USER-side operators -- the counterpart of
the `LinearMap` a Julia user wraps around `ldiv!(factorize(A - sigma*I))` (docs/src/index.md:246-249, 273-287) -- not
part of the library's hot path: they exist so that BASELINE config 4 (ComplexF64 shift-invert) can run with every
vector resident in HBM instead of crossing PCIe twice per product.

    TridiagonalShiftInvert(dl, d, du, sigma)   y = (T - sigma I)^{-1} x  with rocSPARSE's pivoting tridiagonal solver
                                               (rocsparse_[dz]gtsv), on the library's stream.  torch is only plumbing
                                               (device buffers for the three diagonals and the solver's work space).
    sparse_shift_invert(A, sigma)              y = (A - sigma I)^{-1} x for a general sparse A: SuperLU factorisation on
                                               the host (scipy), both triangular solves of every product on the device
                                               through the LIBRARY's own operator `ks_operator_lu` (hand-written
                                               synchronisation-free solve, csrc/ks_sptrsv.hpp) -- no vendor call, no torch.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import api


def _rocsparse():
    for path in ("/opt/rocm/lib/librocsparse.so", "librocsparse.so"):
        try:
            return C.CDLL(path)
        except OSError:
            continue
    raise ImportError("librocsparse.so not found (the tridiagonal shift-invert operator needs rocSPARSE)")


class TridiagonalShiftInvert:
    """Builds `api.Operator` for y = (T - sigma I)^{-1} x, T tridiagonal with sub-/main/super-diagonals dl (n-1), d (n),
    du (n-1).  The eigenvalues lambda of T closest to sigma are sigma + 1/theta for the largest-magnitude theta."""

    def __init__(self, dl, d, du, sigma=0.0, ctx: api.Context | None = None):
        import torch

        self.ctx = ctx or api.default_context()
        cplx = any(np.asarray(a).dtype.kind == "c" for a in (dl, d, du)) or isinstance(sigma, complex)
        self.dtype = np.complex128 if cplx else np.float64
        n = len(d)
        self.n = n
        dev = torch.device("cuda", self.ctx.device)
        L = np.zeros(n, dtype=self.dtype)
        L[1:] = dl                                    # rocSPARSE convention: dl[0] = 0, du[n-1] = 0, all of length n
        U = np.zeros(n, dtype=self.dtype)
        U[:-1] = du
        D = np.asarray(d, dtype=self.dtype) - sigma
        self._dl, self._d, self._du = (torch.as_tensor(a, device=dev) for a in (L, D, U))
        self._lib = _rocsparse()
        self._h = C.c_void_p()
        self._check(self._lib.rocsparse_create_handle(C.byref(self._h)))
        self._check(self._lib.rocsparse_set_stream(self._h, C.c_void_p(self.ctx.stream)))
        pre = "z" if cplx else "d"
        self._solve = getattr(self._lib, f"rocsparse_{pre}gtsv")
        size = C.c_size_t(0)
        probe = torch.zeros(n, dtype=torch.complex128 if cplx else torch.float64, device=dev)
        self._check(getattr(self._lib, f"rocsparse_{pre}gtsv_buffer_size")(
            self._h, C.c_int(n), C.c_int(1), C.c_void_p(self._dl.data_ptr()), C.c_void_p(self._d.data_ptr()),
            C.c_void_p(self._du.data_ptr()), C.c_void_p(probe.data_ptr()), C.c_int(n), C.byref(size)))
        self._buf = torch.empty(max(int(size.value), 16), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)

        def apply(y, x):          # called with two tensors aliasing columns of V, on the library's stream
            y.copy_(x)
            self._check(self._solve(self._h, C.c_int(n), C.c_int(1), C.c_void_p(self._dl.data_ptr()), C.c_void_p(self._d.data_ptr()),
                                    C.c_void_p(self._du.data_ptr()), C.c_void_p(y.data_ptr()), C.c_int(n), C.c_void_p(self._buf.data_ptr())))

        self.operator = api.device_operator(apply, n, self.dtype, self.ctx)
        self.operator._owner = self      # the diagonals and the rocSPARSE handle live as long as the operator

    @staticmethod
    def _check(rc):
        if rc != 0:
            raise RuntimeError(f"rocSPARSE returned status {rc}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rocsparse_destroy_handle(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sparse_shift_invert(A, sigma=0.0, ctx: api.Context | None = None, symmetric_pattern: bool | None = None,
                        residual_limit: float = 1e-10, **splu_kw) -> api.Operator:
    """`api.Operator` for y = (A - sigma I)^{-1} x, A scipy.sparse.  The factorisation runs once on the host
    (`scipy.sparse.linalg.splu`, i.e. SuperLU -- the role SuiteSparse plays behind `factorize` in
    docs/src/index.md:246-249); its triangular factors live in HBM and are applied on the device.  The cost of a product
    is the length of the factors' dependency chains (`operator.lu_info`), so the ordering matters: for matrices with a
    symmetric non-zero pattern (`symmetric_pattern`, detected if None) minimum degree on A + A' with diagonal pivots gives
    far shorter chains and less fill than SuperLU's default column ordering -- kept only if one residual check of the
    factors passes (`residual_limit`), otherwise replaced by the partially pivoted default.  `operator.factor_residual`
    reports the residual of the factors in use."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    n = A.shape[0]
    cplx = A.dtype.kind == "c" or isinstance(sigma, complex)
    M = (sp.csc_matrix(A, dtype=np.complex128 if cplx else np.float64) - sigma * sp.identity(n, format="csc")).tocsc()
    if symmetric_pattern is None:
        P = M.copy()
        P.data[:] = 1.0
        symmetric_pattern = (P != P.T).nnz == 0
    unpivoted = bool(symmetric_pattern and not splu_kw)
    if unpivoted:
        splu_kw = dict(permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    lu = spla.splu(M, **splu_kw)
    if unpivoted:
        # Diagonal-only pivoting is not backward stable: A - sigma I is indefinite and nearly singular exactly when sigma
        # lies inside the spectrum (the use case), and element growth / tiny pivots then go unnoticed.  One solve against
        # a known right-hand side decides: above `residual_limit` the factors are discarded and the default,
        # partially pivoted factorisation (what `lu` / `factorize` of the reference workflow do) is used instead.
        rel = _relative_residual(M, lu)
        if not (rel <= residual_limit):
            import warnings

            warnings.warn(f"sparse_shift_invert: the diagonally pivoted factorisation has relative residual {rel:.1e} "
                          f"(> {residual_limit:.0e}); refactorising with partial pivoting", RuntimeWarning, stacklevel=2)
            lu = spla.splu(M)
    op = api.splu_operator(lu, ctx)
    op.factor_residual = _relative_residual(M, lu)
    return op


def _relative_residual(M, lu) -> float:
    """|| M y - b || / (||M||_1 ||y|| + ||b||) for one deterministic right-hand side (host solve of the same factors)."""
    import scipy.sparse.linalg as spla

    n = M.shape[0]
    b = np.cos(0.7 * np.arange(n) + 0.3).astype(M.dtype)
    y = lu.solve(b)
    if not np.all(np.isfinite(y)):
        return float("inf")
    return float(np.linalg.norm(M @ y - b) / (spla.norm(M, 1) * np.linalg.norm(y) + np.linalg.norm(b)))
