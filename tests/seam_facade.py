"""Test-side stand-in for `HipBasis{T} <: AbstractMatrix{T}` of arnoldimethod.jl_amd/julia/KrylovSchurHIP.jl.

The reference accepts any `AbstractMatrix` as the Krylov basis (`ArnoldiWorkspace(V, H; V_tmp, Q)`,
src/ArnoldiMethod.jl:81-92) and its unmodified `_partialschur` / `orthogonalize!` / `reinitialize!` then reach the basis
only through the handful of verbs SURVEY.md section 8b enumerates.  No Julia runtime exists in the image, so the seam is
replayed with the ORACLE's line-by-line restatement of that code (oracle/arnoldi.py) as the caller: `DeviceBasis` below is
a matrix-like object whose every operation the oracle's code applies to `V` / `V_tmp` forwards to exactly one C entry
point (through ctypes), the way every method of `HipBasis` / `HipColumn` / `HipColumns` forwards to one `ccall`:

    V[:, j], V[:, a:b]                         -> views, no data movement
    np.linalg.norm(v)                          -> ks_col_norm            src/expansion.jl:24,41,48,81,88,96
    v /= s                                     -> ks_col_div             src/expansion.jl:28,56,106
    A.mul_(y, x)                               -> ks_apply               src/expansion.jl:121
    Vprev.conj().T @ v                         -> ks_gemv_t              src/expansion.jl:37,46,84,93
    v -= Vprev @ h                             -> ks_gemv_n_sub          src/expansion.jl:38,47,85,94
    V_tmp[:, a:b] = V[:, a:c] @ Q[a:c, a:b]    -> ks_rotate (in place; V_tmp aliases V)   src/run.jl:363,382
    V[:, a:b] = V_tmp[:, a:b]                  -> nothing (already in place)              src/run.jl:364,383
    V[:, k] = V[:, m]                          -> ks_col_copy            src/run.jl:365
    v[:] = host vector                         -> ks_col_upload          (rand! / copyto!, src/expansion.jl:21, run.jl:126)
    np.asarray(V[:, :k])                       -> ks_cols_download       (result views, src/run.jl:375,389)

Anything else raises: a replay that silently computed on the host would prove nothing.  `calls` counts the verbs.
Test infrastructure only -- nothing in the product imports this."""
from __future__ import annotations

from collections import Counter

import numpy as np


class _Prod:
    """`cols @ M` not yet evaluated (the reference always consumes it by an in-place update or an assignment)."""

    def __init__(self, cols, M):
        self.cols, self.M = cols, np.asarray(M)


class _Adj:
    def __init__(self, cols):
        self.cols = cols

    def __matmul__(self, v):  # Vprev' * v  /  mul!(h, Vprev', v)
        assert isinstance(v, Column) and self.cols.j0 == 0, "only V[:, 0:j]' * V[:, jv] occurs in the reference"
        b = self.cols.basis
        b.calls["gemv_t"] += 1
        return b.ws.gemv_t(self.cols.ncols, v.j)


class _Conj:
    def __init__(self, cols):
        self.cols = cols

    @property
    def T(self):
        return _Adj(self.cols)


class Column:
    """view(V, :, j+1)"""

    def __init__(self, basis, j):
        self.basis, self.j = basis, j
        self.shape = (basis.n,)
        self.dtype = basis.dtype

    def __len__(self):
        return self.basis.n

    def __array_function__(self, func, types, args, kwargs):
        if func is np.linalg.norm and len(args) == 1 and not kwargs:
            self.basis.calls["norm"] += 1
            return self.basis.ws.norm(self.j)
        raise TypeError(f"{func.__name__} on a device column: not a verb of the seam")

    def __array__(self, dtype=None, copy=None):
        raise TypeError("a device column does not convert to a host array implicitly")

    def __itruediv__(self, s):  # v ./= s
        self.basis.calls["div"] += 1
        self.basis.ws.div(self.j, float(s))
        return self

    def __isub__(self, prod):  # mul!(v, Vprev, h, -1, 1)
        assert isinstance(prod, _Prod) and prod.cols.j0 == 0 and prod.M.ndim == 1 and prod.M.shape[0] == prod.cols.ncols
        self.basis.calls["gemv_n_sub"] += 1
        self.basis.ws.gemv_n_sub(prod.cols.ncols, self.j, prod.M)
        return self

    def __setitem__(self, key, value):  # rand!(v) / copyto!(v, v1): the oracle fills through `v[:] = host values`
        assert key == slice(None)
        self.basis.calls["upload"] += 1
        self.basis.ws.set_col(self.j, np.asarray(value, dtype=self.basis.dtype))

    def download(self):
        return self.basis.ws.col(self.j)


class Columns:
    """view(V, :, a:b)"""

    def __init__(self, basis, j0, ncols):
        self.basis, self.j0, self.ncols = basis, j0, ncols
        self.shape = (basis.n, ncols)
        self.dtype = basis.dtype

    def conj(self):
        return _Conj(self)

    def __matmul__(self, M):
        return _Prod(self, M)

    def __getitem__(self, key):
        if key == (slice(None), slice(None)):
            return self
        raise TypeError(key)

    def __setitem__(self, key, value):  # copyto!(view(V, :, a:b), view(V_tmp, :, a:b))
        assert key == (slice(None), slice(None))
        assert isinstance(value, Columns) and (value.j0, value.ncols) == (self.j0, self.ncols) and value.basis.ws is self.basis.ws

    def __array__(self, dtype=None, copy=None):  # Array(decomp.Q)
        self.basis.calls["download"] += 1
        return self.basis.ws.cols(self.j0, self.ncols)

    def __array_function__(self, func, types, args, kwargs):
        raise TypeError(f"{func.__name__} on device columns: not a verb of the seam")


class DeviceBasis:
    """`HipBasis`: the n x (maxdim+1) basis in HBM behind an `ArnoldiWorkspace` handle of the library.  `alias()` gives the
    object passed as `V_tmp` (the rotation is in place; the reference's copy back becomes a no-op)."""

    def __init__(self, ws, calls=None):
        self.ws = ws
        self.n, self.dtype = ws.n, ws.dtype
        self.shape = (ws.n, ws.maxdim + 1)
        self.calls = calls if calls is not None else Counter()

    def alias(self):
        return DeviceBasis(self.ws, self.calls)

    @staticmethod
    def _rng(s, hi):
        a, b, st = s.indices(hi)
        assert st == 1
        return a, max(b - a, 0)

    def __getitem__(self, key):
        rows, cols = key
        assert rows == slice(None)
        if isinstance(cols, (int, np.integer)):
            return Column(self, int(cols))
        a, c = self._rng(cols, self.shape[1])
        return Columns(self, a, c)

    def __setitem__(self, key, value):
        rows, cols = key
        assert rows == slice(None)
        if isinstance(cols, (int, np.integer)):  # copyto!(view(V,:,k+1), view(V,:,maxdim+1))   src/run.jl:365
            assert isinstance(value, Column) and value.basis.ws is self.ws
            self.calls["col_copy"] += 1
            self.ws.copy_col(int(cols), value.j)
            return
        a, r = self._rng(cols, self.shape[1])
        if isinstance(value, _Prod):  # mul!(view(V_tmp,:,a:b), view(V,:,a:c), view(Q,a:c,a:b))   src/run.jl:363,382
            assert value.cols.basis.ws is self.ws and value.cols.j0 == a and value.M.shape == (value.cols.ncols, r)
            if r > 0 and value.cols.ncols > 0:
                self.calls["rotate"] += 1
                self.ws.rotate(a, np.asfortranarray(value.M))
            return
        # copyto!(view(V,:,a:b), view(V_tmp,:,a:b)): already in place
        assert isinstance(value, Columns) and value.basis.ws is self.ws and (value.j0, value.ncols) == (a, r)


class DeviceOperator:
    """`HipOperator`: mul!(y, A, x) on two columns of the basis -> ks_apply."""

    def __init__(self, op, n, dtype):
        self.op, self.shape, self.dtype = op, (n, n), np.dtype(dtype)

    def mul_(self, y, x):
        assert isinstance(y, Column) and isinstance(x, Column) and y.basis.ws is x.basis.ws
        y.basis.calls["apply"] += 1
        y.basis.ws.apply(self.op, x.j, y.j)
