"""ORACLE (test infrastructure): ctypes access to oracle/_build/libkschur_cpuref.so -- the product's
C++ host driver running on an OpenMP CPU backend that issues the reference's un-fused op sequence.

Used by tests (host driver vs the Python oracle, no GPU needed) and by bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libkschur_cpuref.so")
WHICH = {"LM": 0, "LR": 1, "SR": 2, "LI": 3, "SI": 4}


def build():
    subprocess.check_call(["make", "-C", _HERE, "--quiet"])


def lib():
    if not os.path.exists(_LIB):
        build()
    L = C.CDLL(_LIB)
    L.ksref_last_error.restype = C.c_char_p
    return L


def _csr32(A):
    import scipy.sparse as sp

    A = sp.csr_matrix(A)
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A


def partialschur_csr(A, nev, which="LM", tol=None, mindim=None, maxdim=None, restarts=200, v1=None, seed=20240917, spmv_threads=0):
    """Returns dict(Q, R, eigenvalues, history...) from the C++ driver + CPU backend."""
    L = lib()
    rp, ci, A = _csr32(A)
    n = A.shape[0]
    cplx = A.dtype.kind == "c" or (v1 is not None and np.asarray(v1).dtype.kind == "c")
    dt = np.complex128 if cplx else np.float64
    val = np.ascontiguousarray(A.data.astype(dt))
    if tol is None:
        tol = float(np.sqrt(np.finfo(np.float64).eps))
    if mindim is None:
        mindim = min(max(10, nev), n)
    if maxdim is None:
        maxdim = min(max(20, 2 * nev), n)
    H = np.zeros((maxdim + 1, maxdim), dtype=dt, order="F")
    V = np.zeros((n, maxdim + 1), dtype=dt, order="F")
    eig = np.zeros(2 * maxdim)
    hi = np.zeros(8, dtype=np.int32)
    hd = np.zeros(8)
    v1p = None
    if v1 is not None:
        v1c = np.ascontiguousarray(np.asarray(v1, dtype=dt))
        v1p = v1c.ctypes.data_as(C.c_void_p)
    rc = L.ksref_partialschur_csr(
        C.c_int(1 if cplx else 0), C.c_int64(n), rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
        val.ctypes.data_as(C.c_void_p), C.c_int(nev), C.c_int(WHICH[which]), C.c_double(tol), C.c_int(mindim),
        C.c_int(maxdim), C.c_int(restarts), v1p, C.c_uint64(seed), C.c_int(spmv_threads),
        H.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p), eig.ctypes.data_as(C.c_void_p),
        hi.ctypes.data_as(C.c_void_p), hd.ctypes.data_as(C.c_void_p),
    )
    if rc != 0:
        raise ValueError(L.ksref_last_error().decode())
    nconv = int(hi[1])
    return dict(
        Q=V[:, :nconv], R=H[:nconv, :nconv], eigenvalues=(eig[0::2] + 1j * eig[1::2])[:nconv], mvproducts=int(hi[0]),
        nconverged=nconv, converged=bool(hi[2]), nev=int(hi[3]), restarts=int(hi[4]), reorth=int(hi[5]),
        breakdowns=int(hi[6]), t_spmv=hd[0], t_orth=hd[1], t_rot=hd[2], t_host=hd[3], H=H, V=V,
    )


def timed_cycles_csr(A, nev, which, mindim, maxdim, cycles, seed=20240917, spmv_threads=0):
    """Fixed-work timing sample for bench.py (cpu_baseline kind 'port')."""
    L = lib()
    rp, ci, A = _csr32(A)
    n = A.shape[0]
    val = np.ascontiguousarray(A.data.astype(np.float64))
    od = np.zeros(8)
    oi = np.zeros(4, dtype=np.int32)
    rc = L.ksref_timed_cycles_csr(
        C.c_int64(n), rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p),
        C.c_int(nev), C.c_int(WHICH[which]), C.c_int(mindim), C.c_int(maxdim), C.c_int(cycles), C.c_uint64(seed),
        C.c_int(spmv_threads), od.ctypes.data_as(C.c_void_p), oi.ctypes.data_as(C.c_void_p),
    )
    if rc != 0:
        raise RuntimeError("ksref_timed_cycles_csr failed")
    return dict(seconds=od[0], t_spmv=od[1], t_orth=od[2], t_rot=od[3], t_host=od[4], steps=int(oi[0]), reorth=int(oi[1]),
                threads=int(L.ksref_num_threads()))


def stream_triad_gbs(n=1 << 27, reps=4):
    """STREAM-triad bandwidth (GB/s) of the host cores with the OpenMP team the baseline uses."""
    L = lib()
    L.ksref_stream_triad_gbs.restype = C.c_double
    return float(L.ksref_stream_triad_gbs(C.c_int64(n), C.c_int(reps)))
