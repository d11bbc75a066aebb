"""Host-side mirror of the reference's public interface for the accelerated path:

    partialschur(A; v1, nev, which, tol, mindim, maxdim, restarts)      src/run.jl:100-129
    partialschur!(A, arnoldi; start_from, initialize, ...)               src/run.jl:152-179
    partialeigen(P)                                                      src/eigvals.jl:92-95
    ArnoldiWorkspace(n, k) / (v1, k)                                     src/ArnoldiMethod.jl:41-93
    PartialSchur, History, targets LM/LR/SR/LI/SI                        src/ArnoldiMethod.jl:130-137,
                                                                         src/run.jl:217-222, src/targets.jl

Same names, argument meaning and error behaviour (ArgumentError / DimensionMismatch) so that the
parity tests read like the reference's own tests.  Julia is not available in the build image, so
this Python layer (ctypes over the C ABI of include/kschur.h) stands in for the Julia glue that is
shipped, untested, in julia/KrylovSchurHIP.jl.  All n-sized arithmetic happens in libkschur_hip.so
on the GPU; this file only validates arguments and moves small host arrays.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import ArgumentError, DimensionMismatch, check

EPS = float(np.finfo(np.float64).eps)
DEFAULT_SEED = 20240917


# ------------------------------------------------------------------ targets (src/targets.jl:7-32)
class Target:
    name = "LM"

    def __repr__(self):
        return f"{self.name}()"


class LM(Target):
    name = "LM"


class LR(Target):
    name = "LR"


class SR(Target):
    name = "SR"


class LI(Target):
    name = "LI"


class SI(Target):
    name = "SI"


def _which_code(which) -> int:
    """_symbol_to_target, src/run.jl:181-185."""
    if isinstance(which, Target):
        return _lib.WHICH[which.name]
    if isinstance(which, type) and issubclass(which, Target):
        return _lib.WHICH[which.name]
    if isinstance(which, str):
        s = which.lstrip(":")
        if s in _lib.WHICH:
            return _lib.WHICH[s]
    raise ArgumentError(f"Unknown target: {which}")


def vtype(A):
    """src/run.jl:9-12."""
    dt = np.dtype(getattr(A, "dtype", np.float64))
    return np.complex128 if dt.kind == "c" else np.float64


def _dtype_code(dt) -> int:
    return _lib.KS_C64 if np.dtype(dt).kind == "c" else _lib.KS_F64


# ------------------------------------------------------------------ context
class Context:
    """One GPU (+ optionally one rank of a multi-GPU job: RCCL communicator or peer-to-peer regions)."""

    def __init__(self, device: int = 0, rank: int = 0, nranks: int = 1, unique_id: bytes | None = None, p2p: bool = False,
                 hostcomm=None):
        """`hostcomm = (allreduce(buf: np.ndarray) -> None, exchange(peers, sendbufs, recvbufs) -> None)` selects the
        host-staged transport (ks_ctx_create_hostcomm): numpy views of the pinned staging buffers are handed to the
        two callables, which must sum `buf` in place over all ranks / fill `recvbufs[i]` from rank `peers[i]`."""
        L = _lib.load()
        h = C.c_void_p()
        self._keep = ()
        if hostcomm is not None:
            allreduce, exchange = hostcomm
            errs = []

            def _ar(_user, buf, count):
                try:
                    allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                    return 0
                except Exception as e:  # noqa: BLE001 - must not propagate through C
                    errs.append(e)
                    return 1

            def _ex(_user, npeers, peers, sbufs, sbytes, rbufs, rbytes):
                try:
                    pl = [int(peers[i]) for i in range(npeers)]
                    sb = [np.ctypeslib.as_array(C.cast(sbufs[i], C.POINTER(C.c_uint8)), shape=(int(sbytes[i]),)) if sbytes[i] else np.zeros(0, np.uint8)
                          for i in range(npeers)]
                    rb = [np.ctypeslib.as_array(C.cast(rbufs[i], C.POINTER(C.c_uint8)), shape=(int(rbytes[i]),)) if rbytes[i] else np.zeros(0, np.uint8)
                          for i in range(npeers)]
                    exchange(pl, sb, rb)
                    return 0
                except Exception as e:  # noqa: BLE001
                    errs.append(e)
                    return 1

            cb1, cb2 = _lib.HOST_ALLREDUCE_FN(_ar), _lib.HOST_EXCHANGE_FN(_ex)
            self._keep = (cb1, cb2, errs)
            self.comm_errors = errs
            check(L.ks_ctx_create_hostcomm(device, rank, nranks, cb1, cb2, None, C.byref(h)))
        elif p2p:
            check(L.ks_ctx_create_p2p(device, rank, nranks, C.byref(h)))
        elif nranks > 1 or unique_id is not None:
            assert unique_id is not None and len(unique_id) == 128
            buf = C.create_string_buffer(unique_id, 128)
            check(L.ks_ctx_create_dist(device, rank, nranks, buf, C.byref(h)))
        else:
            check(L.ks_ctx_create(device, C.byref(h)))
        self._h = h
        self.device, self.rank, self.nranks = device, rank, nranks

    # peer-to-peer transport: export this rank's region handle / map everybody else's (include/kschur.h)
    def p2p_handle(self) -> bytes:
        buf = C.create_string_buffer(64)
        check(_lib.load().ks_ctx_p2p_handle(self._h, buf))
        return buf.raw

    def p2p_attach(self, handles: list[bytes]):
        assert len(handles) == self.nranks and all(len(x) == 64 for x in handles)
        buf = C.create_string_buffer(b"".join(handles), 64 * self.nranks)
        check(_lib.load().ks_ctx_p2p_attach(self._h, buf))

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(_lib.load().ks_comm_unique_id(buf))
        return buf.raw

    def synchronize(self):
        check(_lib.load().ks_ctx_synchronize(self._h))

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        check(_lib.load().ks_ctx_stream(self._h, C.byref(s)))
        return s.value or 0

    # per-kernel-class HIP-event timing (bench.py)
    PROFILE_CLASSES = ("spmv", "dots", "axpy", "scale", "rotate", "fin", "fused")

    def profile_enable(self, on: bool = True):
        check(_lib.load().ks_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        check(_lib.load().ks_profile_reset(self._h))

    def profile_get(self) -> dict:
        n = len(self.PROFILE_CLASSES)
        ms, by, cnt = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int64)
        check(_lib.load().ks_profile_get(self._h, n, ms.ctypes.data, by.ctypes.data, cnt.ctypes.data))
        return {k: dict(ms=float(ms[i]), bytes=float(by[i]), count=int(cnt[i])) for i, k in enumerate(self.PROFILE_CLASSES)}

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().ks_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ------------------------------------------------------------------ operators
class Operator:
    """Anything with mul!(y, A, x), eltype, size (src/run.jl:21-22)."""

    def __init__(self, ctx, handle, shape, dtype, keep=()):
        self.ctx, self._h, self.shape, self.dtype, self._keep = ctx, handle, shape, np.dtype(dtype), keep

    @property
    def format(self) -> dict:
        """Device layout of a stored matrix: bytes streamed per non-zero and dictionary size (0 = plain CSR)."""
        b, d, lay = C.c_double(), C.c_int(), C.c_int()
        check(_lib.load().ks_operator_format(self._h, C.byref(b), C.byref(d), C.byref(lay)))
        return dict(bytes_per_nnz=b.value, ndict=d.value, layout=_lib.LAYOUTS.get(lay.value, "?"))

    def close(self):
        # never touch a handle whose context is already gone (interpreter shutdown order is arbitrary)
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            _lib.load().ks_operator_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _check_op(rc: int, op):
    """`check(rc)` for calls that may run a Python operator callback: an exception raised inside the callback
    cannot cross the C ABI (the callback returns 1 -> KS_ERR_OPERATOR); re-raise THAT exception here instead
    of the generic library error, chained to it."""
    errs = getattr(op, "errors", None)
    if errs:
        e = errs.pop(0)
        del errs[:]
        try:
            check(rc)
        except Exception as lib_err:  # noqa: BLE001
            raise e from lib_err
        raise e
    check(rc)


def csr_operator(A, ctx: Context | None = None) -> Operator:
    """Device CSR operand from a scipy.sparse matrix (CSR or CSC; CSC is what Julia hands over) or a
    dense ndarray.  Integer/bool matrices are promoted like `vtype` does (test/partial_schur.jl:41-45)."""
    import scipy.sparse as sp

    ctx = ctx or default_context()
    L = _lib.load()
    shp = A.shape
    if len(shp) != 2 or shp[0] != shp[1]:
        raise DimensionMismatch(f"matrix is not square: dimensions are {tuple(shp)}")
    dt = vtype(A)
    if sp.issparse(A):
        if A.format == "csc":
            M, layout = A, _lib.KS_CSC
        else:
            M, layout = A.tocsr(), _lib.KS_CSR
    else:
        M, layout = sp.csr_matrix(np.asarray(A)), _lib.KS_CSR
    ptr = np.ascontiguousarray(M.indptr)
    idx = np.ascontiguousarray(M.indices)
    if ptr.dtype != idx.dtype:
        ptr = ptr.astype(np.int64)
        idx = idx.astype(np.int64)
    itype = _lib.KS_I64 if ptr.dtype == np.int64 else _lib.KS_I32
    if ptr.dtype not in (np.int32, np.int64):
        ptr, idx, itype = ptr.astype(np.int64), idx.astype(np.int64), _lib.KS_I64
    val = np.ascontiguousarray(M.data.astype(dt))
    h = C.c_void_p()
    check(
        L.ks_operator_csr(
            ctx._h, shp[0], shp[1], M.nnz, ptr.ctypes.data, idx.ctypes.data, val.ctypes.data if M.nnz else None,
            layout, 0, itype, _dtype_code(dt), C.byref(h),
        )
    )
    return Operator(ctx, h, tuple(shp), dt)


def lu_operator(L, U, perm_in=None, perm_out=None, scale=None, ctx: Context | None = None) -> Operator:
    """y = P_out U^-1 L^-1 P_in (scale o x) with both sparse triangular solves on the device (`ks_operator_lu`): the
    `ldiv!(y, F, x)` of the LinearMap a user of the reference wraps around a host factorisation
    (docs/src/index.md:246-249).  `L`, `U`: scipy.sparse triangular factors (any format; converted to CSR here),
    `perm_in[i]` = the entry of x that becomes row i of the triangular system, `perm_out[i]` = the entry of y that
    receives solution entry i.  See `splu_operator` for the SuperLU convention."""
    import scipy.sparse as sp

    ctx = ctx or default_context()
    lib = _lib.load()
    n = L.shape[0]
    if L.shape != (n, n) or U.shape != (n, n):
        raise DimensionMismatch(f"factors are not square of one size: {L.shape}, {U.shape}")
    dt = np.dtype(np.complex128 if (L.dtype.kind == "c" or U.dtype.kind == "c") else np.float64)
    arrs = []
    for M in (L, U):
        M = sp.csr_matrix(M)
        M.sort_indices()
        arrs += [np.ascontiguousarray(M.indptr, dtype=np.int64), np.ascontiguousarray(M.indices, dtype=np.int32),
                 np.ascontiguousarray(M.data, dtype=dt)]
    pin = None if perm_in is None else np.ascontiguousarray(perm_in, dtype=np.int32)
    pout = None if perm_out is None else np.ascontiguousarray(perm_out, dtype=np.int32)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
    for a, name in ((pin, "perm_in"), (pout, "perm_out"), (sc, "scale")):
        if a is not None and a.shape != (n,):
            raise DimensionMismatch(f"{name} must have {n} entries")
    ptr = lambda a: None if a is None or a.size == 0 else a.ctypes.data  # noqa: E731
    h = C.c_void_p()
    check(lib.ks_operator_lu(ctx._h, n, _dtype_code(dt), arrs[0].ctypes.data, ptr(arrs[1]), ptr(arrs[2]), arrs[3].ctypes.data,
                             ptr(arrs[4]), ptr(arrs[5]), ptr(pin), ptr(pout), ptr(sc), C.byref(h)))
    op = Operator(ctx, h, (n, n), dt)
    v = [C.c_int64() for _ in range(4)]
    check(lib.ks_operator_lu_info(h, *[C.byref(x) for x in v]))
    op.lu_info = dict(nnz_l=v[0].value, nnz_u=v[1].value, levels_l=v[2].value, levels_u=v[3].value)
    for upper, tag in ((0, "l"), (1, "u")):
        w = [C.c_int64() for _ in range(3)]
        g = C.c_int()
        check(lib.ks_operator_lu_layout(h, upper, C.byref(w[0]), C.byref(w[1]), C.byref(w[2]), C.byref(g)))
        op.lu_info.update({f"rows_{tag}": w[0].value, f"run_rows_{tag}": w[1].value, f"top_rows_{tag}": w[2].value, f"groups_{tag}": g.value})
    return op


def splu_operator(lu, ctx: Context | None = None) -> Operator:
    """The device operator x -> A^-1 x from a `scipy.sparse.linalg.splu` factorisation (`Pr A Pc = L U`:
    z[perm_r] = x, L U w = z, y = w[perm_c])."""
    n = lu.shape[0]
    inv_r = np.empty(n, dtype=np.int32)
    inv_r[lu.perm_r] = np.arange(n, dtype=np.int32)
    inv_c = np.empty(n, dtype=np.int32)
    inv_c[lu.perm_c] = np.arange(n, dtype=np.int32)
    return lu_operator(lu.L, lu.U, perm_in=inv_r, perm_out=inv_c, ctx=ctx)


def host_operator(fn, n: int, dtype=np.float64, ctx: Context | None = None) -> Operator:
    """Opaque host operator: `fn(y, x)` fills y = A*x on numpy views (a LinearMap wrapping ldiv!,
    docs/src/index.md:246-249).  Columns are staged over PCIe by the library."""
    ctx = ctx or default_context()
    L = _lib.load()
    dt = np.dtype(vtype(np.empty(0, dtype=dtype)))
    err = []

    def _cb(_user, xp, yp):
        try:
            x = np.ctypeslib.as_array(C.cast(xp, C.POINTER(C.c_double)), shape=(n * (2 if dt.kind == "c" else 1),)).view(dt)
            y = np.ctypeslib.as_array(C.cast(yp, C.POINTER(C.c_double)), shape=(n * (2 if dt.kind == "c" else 1),)).view(dt)
            fn(y, x)
            return 0
        except Exception as e:  # noqa: BLE001 - must not propagate through C
            err.append(e)
            return 1

    cb = _lib.HOST_APPLY_FN(_cb)
    h = C.c_void_p()
    check(L.ks_operator_host_callback(ctx._h, n, _dtype_code(dt), cb, None, C.byref(h)))
    op = Operator(ctx, h, (n, n), dt, keep=(cb, err))
    op.errors = err
    return op


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can wrap a raw device pointer without copying."""

    def __init__(self, ptr: int, n: int, dt: np.dtype):
        self.__cuda_array_interface__ = {
            "shape": (n,), "typestr": "<c16" if dt.kind == "c" else "<f8", "data": (ptr, False), "version": 2, "strides": None,
        }


def device_operator(fn, n: int, dtype=np.float64, ctx: Context | None = None) -> Operator:
    """Opaque DEVICE operator: `fn(y, x)` receives two torch tensors that alias columns of V in HBM and must
    enqueue y = A*x on the current torch stream (which is the library's stream while `fn` runs).  This is the
    device-resident form of the `mul!(y, A, x)` seam (src/expansion.jl:121): e.g. a dense matrix via
    `torch.mv`, a matrix-free stencil, a preconditioned solve.  torch is only plumbing here."""
    import torch

    ctx = ctx or default_context()
    L = _lib.load()
    dt = np.dtype(vtype(np.empty(0, dtype=dtype)))
    err = []

    def _cb(_user, xp, yp, stream):
        try:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=torch.device("cuda", ctx.device))):
                x = torch.as_tensor(_DevArray(xp, n, dt), device=torch.device("cuda", ctx.device))
                y = torch.as_tensor(_DevArray(yp, n, dt), device=torch.device("cuda", ctx.device))
                fn(y, x)
            return 0
        except Exception as e:  # noqa: BLE001 - must not propagate through C
            err.append(e)
            return 1

    cb = _lib.DEVICE_APPLY_FN(_cb)
    h = C.c_void_p()
    check(L.ks_operator_device_callback(ctx._h, n, _dtype_code(dt), cb, None, C.byref(h)))
    op = Operator(ctx, h, (n, n), dt, keep=(cb, err))
    op.errors = err
    return op


def dense_operator(A, ctx: Context | None = None) -> Operator:
    """mul!(y, A::Matrix, x): the dense matrix lives row-major in HBM (ks_operator_dense)."""
    ctx = ctx or default_context()
    A = np.asarray(A)
    if A.ndim != 2 or A.shape[0] != A.shape[1]:
        raise DimensionMismatch(f"matrix is not square: dimensions are {A.shape}")
    dt = vtype(A)
    col_major = A.flags.f_contiguous and not A.flags.c_contiguous
    M = np.asfortranarray(A, dtype=dt) if col_major else np.ascontiguousarray(A, dtype=dt)
    h = C.c_void_p()
    check(_lib.load().ks_operator_dense(ctx._h, A.shape[0], M.ctypes.data, A.shape[0], 1 if col_major else 0, _dtype_code(dt), C.byref(h)))
    return Operator(ctx, h, A.shape, dt)


def as_operator(A, ctx: Context | None = None) -> Operator:
    import scipy.sparse as sp

    if isinstance(A, Operator):
        return A
    if isinstance(A, np.ndarray):
        # a mostly-zero array is cheaper as CSR (12 B per stored entry vs 8 B per entry of the dense stream)
        if A.ndim == 2 and A.size and np.count_nonzero(A) > 0.5 * A.size:
            return dense_operator(A, ctx)
        return csr_operator(A, ctx)
    if sp.issparse(A):
        return csr_operator(A, ctx)
    if all(hasattr(A, k) for k in ("L", "U", "perm_r", "perm_c", "solve")):
        return splu_operator(A, ctx)  # a scipy SuperLU factorisation: the operator is x -> A^-1 x, applied on the device
    shp = getattr(A, "shape", None)
    if shp is None or len(shp) != 2 or shp[0] != shp[1]:
        raise DimensionMismatch(f"matrix is not square: dimensions are {shp}")
    if hasattr(A, "mul_"):
        return host_operator(lambda y, x: A.mul_(y, x), shp[0], getattr(A, "dtype", np.float64), ctx)
    if hasattr(A, "matvec"):
        def f(y, x):
            y[:] = A.matvec(x)
        return host_operator(f, shp[0], getattr(A, "dtype", np.float64), ctx)
    raise TypeError("operator must be a scipy.sparse matrix, an ndarray, an Operator, or implement mul_/matvec")


# ------------------------------------------------------------------ workspace (src/ArnoldiMethod.jl:41-93)
class ArnoldiWorkspace:
    """V (n x (k+1), in HBM), H ((k+1) x k, host, zero-initialised), Q (k x k, host).

    ArnoldiWorkspace(n, k [, dtype])  or  ArnoldiWorkspace(v1, k)  (the array type follows v1).

    Deviation from the reference, on purpose: `ArnoldiWorkspace(v1, k)` (src/ArnoldiMethod.jl:71-79) only takes the ARRAY
    TYPE from `v1` and `partialschur!` then starts from `rand!` (src/run.jl:177); here the workspace remembers `v1` and
    `partialschur_(A, ws)` with `initialize=True, start_from=1` starts from it -- the tests need reproducible start
    vectors and Python has no array-type dispatch to express the original meaning.  Pass `ArnoldiWorkspace(n, k)` to get
    the reference's random start (counter-based RNG, `set_seed`)."""

    def __init__(self, n_or_v1, krylov_dimension: int, dtype=np.float64, ctx: Context | None = None,
                 n_global: int | None = None, row_begin: int = 0):
        self.ctx = ctx or default_context()
        L = _lib.load()
        self._v1 = None
        if isinstance(n_or_v1, (int, np.integer)):
            n = int(n_or_v1)
        else:
            v1 = np.asarray(n_or_v1)
            n = v1.shape[0]
            dtype = vtype(v1)
            self._v1 = np.ascontiguousarray(v1.astype(dtype))
        self.dtype = np.dtype(vtype(np.empty(0, dtype=dtype)))
        ng = n if n_global is None else int(n_global)
        if not krylov_dimension <= ng:
            raise ArgumentError("Krylov dimension should be less than matrix order.")
        h = C.c_void_p()
        check(L.ks_workspace_create(self.ctx._h, n, ng, row_begin, krylov_dimension, _dtype_code(self.dtype), C.byref(h)))
        self._h = h
        self.n, self.n_global, self.maxdim = n, ng, krylov_dimension
        self.H = self._host_matrix(L.ks_workspace_H, krylov_dimension + 1, krylov_dimension)
        self.Q = self._host_matrix(L.ks_workspace_Q, krylov_dimension, krylov_dimension)

    def _host_matrix(self, getter, rows, cols):
        p, ld = C.c_void_p(), C.c_int()
        check(getter(self._h, C.byref(p), C.byref(ld)))
        nd = 2 if self.dtype.kind == "c" else 1
        flat = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(ld.value * cols * nd,))
        return flat.view(self.dtype).reshape((ld.value, cols), order="F")[:rows, :]

    # --- the verbs the reference applies to V ---
    def set_seed(self, seed: int):
        check(_lib.load().ks_workspace_set_seed(self._h, seed))

    def col(self, j: int) -> np.ndarray:
        out = np.empty(self.n, dtype=self.dtype)
        check(_lib.load().ks_col_download(self._h, j, out.ctypes.data))
        return out

    def cols(self, j0: int, ncols: int) -> np.ndarray:
        out = np.empty((self.n, ncols), dtype=self.dtype, order="F")
        if ncols and self.n:
            check(_lib.load().ks_cols_download(self._h, j0, ncols, out.ctypes.data, self.n))
        return out

    def set_col(self, j: int, v):
        v = np.ascontiguousarray(np.asarray(v, dtype=self.dtype))
        if v.shape != (self.n,):
            raise ArgumentError("v1 should have the same dimension as A")
        check(_lib.load().ks_col_upload(self._h, j, v.ctypes.data))

    def set_cols(self, j0: int, M):
        M = np.asfortranarray(np.asarray(M, dtype=self.dtype))
        check(_lib.load().ks_cols_upload(self._h, j0, M.shape[1], M.ctypes.data, M.shape[0]))

    @property
    def V(self) -> np.ndarray:
        """Host copy of the whole basis (tests / small problems only)."""
        return self.cols(0, self.maxdim + 1)

    def fill_uniform(self, j: int, seed: int):
        check(_lib.load().ks_col_fill_uniform(self._h, j, seed))

    def norm(self, j: int) -> float:
        out = C.c_double()
        check(_lib.load().ks_col_norm(self._h, j, C.byref(out)))
        return out.value

    def div(self, j: int, s: float):
        check(_lib.load().ks_col_div(self._h, j, s))

    def copy_col(self, dst: int, src: int):
        check(_lib.load().ks_col_copy(self._h, dst, src))

    def apply(self, A: Operator, jsrc: int, jdst: int):
        _check_op(_lib.load().ks_apply(A._h, self._h, jsrc, jdst), A)

    def gemv_t(self, j: int, jv: int) -> np.ndarray:
        h = np.empty(j, dtype=self.dtype)
        check(_lib.load().ks_gemv_t(self._h, j, jv, h.ctypes.data))
        return h

    def gemv_n_sub(self, j: int, jv: int, h):
        h = np.ascontiguousarray(np.asarray(h, dtype=self.dtype))
        check(_lib.load().ks_gemv_n_sub(self._h, j, jv, h.ctypes.data))

    def rotate(self, c0: int, Qblock):
        Qb = np.asfortranarray(np.asarray(Qblock, dtype=self.dtype))
        c, r = Qb.shape
        check(_lib.load().ks_rotate(self._h, c0, c, r, Qb.ctypes.data, c))

    def basis_times(self, c: int, Y) -> np.ndarray:
        Y = np.asarray(Y)
        ydt = np.complex128 if (Y.dtype.kind == "c" or self.dtype.kind == "c") else np.float64
        Yf = np.asfortranarray(Y.astype(ydt))
        out = np.empty((self.n, Yf.shape[1]), dtype=ydt, order="F")
        check(_lib.load().ks_basis_times(self._h, c, Yf.shape[1], Yf.ctypes.data, Yf.shape[0], _dtype_code(ydt), out.ctypes.data, self.n))
        return out

    def orthogonalize(self, j: int) -> bool:
        ok = C.c_int()
        check(_lib.load().ks_orthogonalize(self._h, j, C.byref(ok)))
        return bool(ok.value)

    def reinitialize(self, j: int = 0, v1=None) -> bool:
        ok = C.c_int()
        p = None
        if v1 is not None:
            v1 = np.ascontiguousarray(np.asarray(v1, dtype=self.dtype))
            p = v1.ctypes.data
        check(_lib.load().ks_reinitialize(self._h, j, p, C.byref(ok)))
        return bool(ok.value)

    def iterate_arnoldi(self, A: Operator, frm: int, to: int):
        st = _lib.ks_expand_stats()
        _check_op(_lib.load().ks_iterate_arnoldi(A._h, self._h, frm, to, C.byref(st)), A)
        return dict(steps=st.steps, reorth=st.reorth, breakdowns=st.breakdowns, explicit_steps=st.explicit_steps)

    def restart(self, active: int, nev: int, which="LM", tol=None, mindim=None, maxdim=None):
        """One Krylov-Schur restart (src/run.jl:278-365): host Schur / grouping / restore, then the
        basis rotation on the device.  `active` is 0-based.  Returns dict(k, nlock, purge, ...)."""
        maxdim = self.maxdim if maxdim is None else maxdim
        mindim = min(max(10, nev), self.n_global) if mindim is None else mindim
        tol = math.sqrt(EPS) if tol is None else tol
        p = _lib.ks_params(nev, _which_code(which), float(tol), mindim, maxdim, 1, 1, 0, 0)
        k, nlock, purge = C.c_int(), C.c_int(), C.c_int()
        lams, rs, groups = np.zeros(2 * maxdim), np.zeros(maxdim), np.zeros(maxdim, dtype=np.int32)
        check(_lib.load().ks_restart(self._h, C.byref(p), active, C.byref(k), C.byref(nlock), C.byref(purge),
                                     lams.ctypes.data, rs.ctypes.data, groups.ctypes.data))
        return dict(k=k.value, nlock=nlock.value, purge=purge.value, eigenvalues=lams[0::2] + 1j * lams[1::2],
                    residuals=rs, groups=groups)

    def expand_restart(self, A: Operator, k: int, active: int, nev: int, which="LM", tol=None, mindim=None, maxdim=None):
        """One whole cycle of `_partialschur`'s loop (src/run.jl:272-365): iterate_arnoldi!(k+1 : maxdim) and the restart in
        ONE library call (what `partialschur` does internally; bit-identical to iterate_arnoldi + restart).  With the explicit
        second pass (`passes == 3`) the early part of the restart's host work overlaps the tail of the expansion; with the
        implicit second pass (default) H is final only when the batch ends.  `k` = basis size the previous restart
        left.  Returns restart()'s dict plus steps / reorth / breakdowns and seconds = (expansion, host, rotation enqueue)."""
        maxdim = self.maxdim if maxdim is None else maxdim
        mindim = min(max(10, nev), self.n_global) if mindim is None else mindim
        tol = math.sqrt(EPS) if tol is None else tol
        p = _lib.ks_params(nev, _which_code(which), float(tol), mindim, maxdim, 1, 1, 0, 0)
        ko, nlock, purge = C.c_int(), C.c_int(), C.c_int()
        lams, rs, groups = np.zeros(2 * maxdim), np.zeros(maxdim), np.zeros(maxdim, dtype=np.int32)
        st = _lib.ks_expand_stats()
        sec = np.zeros(3)
        _check_op(_lib.load().ks_expand_restart(A._h, self._h, C.byref(p), active, k, C.byref(ko), C.byref(nlock), C.byref(purge),
                                                lams.ctypes.data, rs.ctypes.data, groups.ctypes.data, C.byref(st), sec.ctypes.data), A)
        return dict(k=ko.value, nlock=nlock.value, purge=purge.value, eigenvalues=lams[0::2] + 1j * lams[1::2], residuals=rs,
                    groups=groups, steps=st.steps, reorth=st.reorth, breakdowns=st.breakdowns, explicit_steps=st.explicit_steps,
                    seconds=tuple(sec))

    def residual_norms(self, A: Operator, ncols: int):
        r, o = C.c_double(), C.c_double()
        _check_op(_lib.load().ks_residual_norms(A._h, self._h, ncols, C.byref(r), C.byref(o)), A)
        return r.value, o.value

    def arnoldi_relation(self, A: Operator, k: int):
        r, o = C.c_double(), C.c_double()
        _check_op(_lib.load().ks_arnoldi_relation(A._h, self._h, k, C.byref(r), C.byref(o)), A)
        return r.value, o.value

    @property
    def passes(self) -> int:
        """Reads of the basis per fused expansion step: 2 (implicit second pass, default) or 3 (KS_PASSES=3)."""
        k = C.c_int()
        check(_lib.load().ks_workspace_passes(self._h, C.byref(k)))
        return k.value

    def set_passes(self, passes: int, max_ratio: float = float("nan")):
        """Per-workspace switch: 2 = implicit second DGKS pass, 3 = explicit; `max_ratio` = largest ||c|| / beta the
        implicit form carries before a step is redone explicitly (NaN keeps the current value, <= 0 removes the limit)."""
        check(_lib.load().ks_workspace_set_passes(self._h, int(passes), float(max_ratio)))

    def set_sstep(self, s: int, pivot_min: float = float("nan"), gram_dev_max: float = float("nan")):
        """s-step (block) expansion: s >= 2 takes the steps of an expansion in blocks of up to s (two reads of the basis per
        BLOCK instead of per step; include/kschur.h, ks_workspace_set_sstep); 0 switches it off.  `pivot_min`: smallest
        Cholesky pivot ratio, `gram_dev_max`: largest deviation of the written block's Gram matrix from I a block may have
        before it is abandoned and redone step by step (NaN keeps the value)."""
        check(_lib.load().ks_workspace_set_sstep(self._h, int(s), float(pivot_min), float(gram_dev_max)))

    @property
    def sstep_info(self) -> dict:
        """Block size in force, blocks completed / abandoned since creation, and of the last batch: smallest pivot ratios of
        the two Gram-Schmidt stages and the largest entry of |Gram matrix of the written block - I|."""
        s, b, a = C.c_int(), C.c_int(), C.c_int()
        d = (C.c_double * 3)()
        check(_lib.load().ks_workspace_sstep_info(self._h, C.byref(s), C.byref(b), C.byref(a), d))
        fr, sa, sd = C.c_int(), C.c_int(), C.c_int()
        check(_lib.load().ks_workspace_fused_rotations(self._h, C.byref(fr), C.byref(sa), C.byref(sd)))
        sr = C.c_int()
        check(_lib.load().ks_workspace_split_rotations(self._h, C.byref(sr)))
        db, dc = C.c_int(), C.c_int()
        check(_lib.load().ks_workspace_deflated_blocks(self._h, C.byref(db), C.byref(dc)))
        return dict(s=s.value, blocks=b.value, abandoned=a.value, pivot_stage1=d[0], pivot_stage2=d[1], gram_dev=d[2], fused_rotations=fr.value,
                    split_rotations=sr.value, chains_adopted=sa.value, chains_dropped=sd.value, deflated_blocks=db.value, deflated_columns=dc.value)

    @property
    def relation_info(self) -> dict:
        """Restarts that cut through a 2 x 2 block of the real Schur form (ks_workspace_relation_info): count and the largest
        dropped entry relative to ||H||_F.  After the first one the s-step expansion stays off for the run.  `probes`: relation
        measurements taken on factorisations a caller vouched for (ks_workspace_relation_probes)."""
        b, w = C.c_int(), C.c_double()
        check(_lib.load().ks_workspace_relation_info(self._h, C.byref(b), C.byref(w)))
        pr = C.c_int()
        check(_lib.load().ks_workspace_relation_probes(self._h, C.byref(pr)))
        return dict(breaks=b.value, worst_leak=w.value, probes=pr.value)

    def assert_arnoldi(self, k: int):
        """The caller vouches that columns 0..k are orthonormal and satisfy, with the H now in `self.H`, the Arnoldi
        relation of k steps (a restart the host language ran itself): re-enables the implicit second pass."""
        check(_lib.load().ks_workspace_assert_arnoldi(self._h, int(k)))

    @property
    def provenance(self) -> int:
        """Steps of the factorisation the library trusts as its own (-1: none) -- see ks_workspace_assert_arnoldi."""
        k = C.c_int()
        check(_lib.load().ks_workspace_provenance(self._h, C.byref(k)))
        return k.value

    @property
    def placement(self) -> dict:
        """Outcome of the placement search at creation: candidates timed, calibration ms of the kept / slowest one."""
        k, b, w, r = C.c_int(), C.c_double(), C.c_double(), C.c_int()
        check(_lib.load().ks_workspace_placement(self._h, C.byref(k), C.byref(b), C.byref(w), C.byref(r)))
        return dict(candidates=k.value, kept_ms=b.value, slowest_ms=w.value, refused=r.value)

    def guard_intact(self) -> bool:
        """KS_GUARD=1 debugging: True unless a kernel wrote outside the basis."""
        ok = C.c_int()
        check(_lib.load().ks_workspace_check_guard(self._h, C.byref(ok)))
        return bool(ok.value)

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            _lib.load().ks_workspace_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sstep_partition(dtype, k0: int, count: int, smax: int) -> list:
    """Block sizes the s-step expansion uses for `count` steps on top of `k0` existing columns (ks_sstep_partition: the
    library's own blk_partition, csrc/ks_block.hpp -- the sizes depend on which kernel forms are switched on; [] = not
    block-capable).  For byte accounting in benchmarks: a block of s steps on k columns reads 8 n (k + s) + 8 n (k + s) and
    writes 8 n s bytes (x2 for ComplexF64) next to its s operator products."""
    if k0 < 1 or count < 1 or smax < 1:
        return []
    cap = 64
    out = (C.c_int * cap)()
    nb = C.c_int(0)
    check(_lib.load().ks_sstep_partition(_dtype_code(dtype), int(k0), int(count), int(smax), out, cap, C.byref(nb)))
    return [int(out[i]) for i in range(min(nb.value, cap))]


# ------------------------------------------------------------------ results
class PartialSchur:
    """src/ArnoldiMethod.jl:130-137.  `Q` stays in HBM (a view of the workspace's V, src/run.jl:375,389)
    and is downloaded on first access; `R` is a view of the workspace's host H."""

    def __init__(self, ws: ArnoldiWorkspace, nconv: int, eigenvalues: np.ndarray):
        self.workspace = ws
        self.nconverged = nconv
        self.R = ws.H[:nconv, :nconv]
        self.eigenvalues = eigenvalues
        self._Q = None

    @property
    def Q(self) -> np.ndarray:
        if self._Q is None:
            self._Q = self.workspace.cols(0, self.nconverged)
        return self._Q

    def __repr__(self):  # src/show.jl:23-33
        return (
            f"PartialSchur decomposition ({'ComplexF64' if self.workspace.dtype.kind == 'c' else 'Float64'}) "
            f"of dimension {self.nconverged}\neigenvalues:\n{self.eigenvalues!r}"
        )


@dataclass
class History:
    """src/run.jl:217-222 (+ timing diagnostics of the device path)."""

    mvproducts: int
    nconverged: int
    converged: bool
    nev: int
    restarts: int = 0
    reorth: int = 0
    breakdowns: int = 0
    explicit_steps: int = 0  # steps the implicit second DGKS pass handed back to the explicit form
    seconds_expand: float = 0.0
    seconds_host: float = 0.0
    seconds_rotate: float = 0.0

    def __str__(self):  # src/show.jl:3-21
        head = "Converged" if self.converged else "Not converged"
        return f"{head}: {self.nconverged} of {self.nev} eigenvalues in {self.mvproducts} matrix-vector products"


# ------------------------------------------------------------------ drivers
def _run(op: Operator, ws: ArnoldiWorkspace, nev, which, tol, mindim, maxdim, restarts, start_from, initialize, v1):
    L = _lib.load()
    p = _lib.ks_params(nev, _which_code(which), float(tol), mindim, maxdim, restarts, start_from, 1 if initialize else 0, 0)
    h = _lib.ks_history()
    eig = np.zeros(2 * max(maxdim, 1))
    v1p = None
    if v1 is not None:
        v1 = np.ascontiguousarray(np.asarray(v1, dtype=ws.dtype))
        v1p = v1.ctypes.data
    _check_op(L.ks_partialschur(op._h, ws._h, C.byref(p), v1p, eig.ctypes.data, C.byref(h)), op)
    lam = (eig[0::2] + 1j * eig[1::2])[: h.nconverged].copy()
    hist = History(h.mvproducts, h.nconverged, bool(h.converged), h.nev, h.restarts, h.reorth, h.breakdowns, h.explicit_steps,
                   h.seconds_expand, h.seconds_host, h.seconds_rotate)
    return PartialSchur(ws, h.nconverged, lam), hist


def partialschur(A, v1=None, nev=None, which="LM", tol=None, mindim=None, maxdim=None, restarts=200,
                 seed=DEFAULT_SEED, ctx: Context | None = None):
    """partialschur(A; v1, nev, which, tol, mindim, maxdim, restarts) -> (PartialSchur, History)

    src/run.jl:100-129.  A: scipy.sparse matrix / ndarray (uploaded as device CSR), an `Operator`, or
    any object with `mul_(y, x)` / `matvec(x)` (host-callback operator)."""
    shp = getattr(A, "shape", None)
    if shp is None or len(shp) != 2 or shp[0] != shp[1]:
        raise DimensionMismatch(f"matrix is not square: dimensions are {tuple(shp) if shp else shp}")
    n = shp[0]
    if nev is None:
        nev = min(6, n)
    if tol is None:
        tol = math.sqrt(EPS)
    if mindim is None:
        mindim = min(max(10, nev), n)
    if maxdim is None:
        maxdim = min(max(20, 2 * nev), n)
    if nev < 1:
        raise ArgumentError("nev cannot be less than 1")
    if not (nev <= mindim <= maxdim <= n):
        raise ArgumentError(f"nev ≤ mindim ≤ maxdim ≤ size(A, 1) does not hold, got {nev} ≤ {mindim} ≤ {maxdim} ≤ {n}")
    _which_code(which)
    if v1 is not None and len(v1) != n:
        raise ArgumentError("v1 should have the same dimension as A")
    v1_complex = v1 is not None and np.asarray(v1).dtype.kind == "c"
    if v1_complex and not isinstance(A, Operator) and np.dtype(getattr(A, "dtype", np.float64)).kind != "c" and hasattr(A, "astype"):
        # ArnoldiWorkspace(v1, maxdim) takes the element type of v1 (src/ArnoldiMethod.jl:71-79): a real matrix
        # with a complex start vector runs in complex arithmetic
        A = A.astype(np.complex128)
    op = as_operator(A, ctx)
    dtype = np.complex128 if (op.dtype.kind == "c" or v1_complex) else np.float64
    if np.dtype(dtype) != op.dtype:
        raise ArgumentError("a complex start vector needs a complex operator")
    ws = ArnoldiWorkspace(n, maxdim, dtype, ctx=op.ctx)
    ws.set_seed(seed)
    return _run(op, ws, nev, which, tol, mindim, maxdim, restarts, 1, True, v1)


def partialschur_(A, arnoldi: ArnoldiWorkspace, start_from=1, initialize=None, nev=None, which="LM", tol=None,
                  mindim=None, maxdim=None, restarts=200):
    """`partialschur!(A, arnoldi; start_from, initialize, ...)`, src/run.jl:152-179."""
    shp = getattr(A, "shape", None)
    if shp is None or len(shp) != 2 or shp[0] != shp[1]:
        raise DimensionMismatch(f"matrix is not square: dimensions are {tuple(shp) if shp else shp}")
    s = shp[0]
    ncolsV = arnoldi.maxdim + 1
    if initialize is None:
        initialize = start_from == 1
    if nev is None:
        nev = min(6, s)
    if tol is None:
        tol = math.sqrt(EPS)
    if mindim is None:
        mindim = min(max(10, nev), s, ncolsV - 1)
    if maxdim is None:
        maxdim = min(max(20, 2 * nev), s, ncolsV - 1)
    if nev < 1:
        raise ArgumentError("nev cannot be less than 1")
    if not (nev <= mindim <= maxdim <= s):
        raise ArgumentError(f"nev ≤ mindim ≤ maxdim ≤ size(A, 1) does not hold, got {nev} ≤ {mindim} ≤ {maxdim} ≤ {s}")
    if not maxdim < ncolsV:
        raise ArgumentError("maxdim should be strictly less than size(arnoldi.V, 2)")
    if not (1 <= start_from <= maxdim):
        raise ArgumentError("start_from should be between 1 and maxdim")
    _which_code(which)
    op = as_operator(A, arnoldi.ctx)
    v1 = arnoldi._v1 if (initialize and start_from == 1) else None
    return _run(op, arnoldi, nev, which, tol, mindim, maxdim, restarts, start_from, initialize, v1)


def partialeigen(P: PartialSchur):
    """partialeigen(P) -> (eigenvalues, eigenvectors), src/eigvals.jl:92-95: LAPACK `eigen(R)` on the
    host (nev x nev) and the tall-skinny product Q*vecs on the device."""
    import scipy.linalg as sla

    k = P.nconverged
    if k == 0:
        return np.zeros(0, dtype=np.complex128), np.zeros((P.workspace.n, 0), dtype=np.complex128)
    vals, vecs = sla.eig(np.array(P.R))
    if P.workspace.dtype.kind == "f" and np.all(vals.imag == 0):
        vecs = vecs.real
    return vals, P.workspace.basis_times(k, vecs)
