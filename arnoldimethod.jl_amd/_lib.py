"""ctypes binding of libkschur_hip.so (include/kschur.h).  No torch types cross this boundary.

The product path has NO CPU fallback: if the HIP library is missing this module raises at load
time, and every compute entry point fails with KS_ERR_NO_DEVICE / KS_ERR_HIP when no gfx950 device
is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkschur_hip.so")

KS_OK, KS_ERR_ARGUMENT, KS_ERR_DIMENSION, KS_ERR_HIP, KS_ERR_RCCL, KS_ERR_QR, KS_ERR_INTERNAL, KS_ERR_NO_DEVICE, KS_ERR_OPERATOR, KS_ERR_COMM = range(10)
KS_F64, KS_C64 = 0, 1
KS_I32, KS_I64 = 0, 1
KS_CSR, KS_CSC = 0, 1
LAYOUTS = {-1: "none", 0: "csr", 1: "csr-vi", 2: "csr-dvi", 3: "sell", 4: "sell-vi", 5: "stencil", 6: "csr-cb"}
WHICH = {"LM": 0, "LR": 1, "SR": 2, "LI": 3, "SI": 4}


class ArgumentError(ValueError):
    """Julia's ArgumentError (src/run.jl:111-116,123-124,165-174,185; src/ArnoldiMethod.jl:62-63,87-90)."""


class DimensionMismatch(ValueError):
    """Julia's DimensionMismatch (checksquare, src/run.jl:110)."""


class QRDidNotConverge(RuntimeError):
    """"QR algorithm did not converge" (src/schurfact.jl:406)."""


class HipError(RuntimeError):
    pass


class CommTimeout(HipError):
    """A peer rank did not take part in an exchange within KS_P2P_TIMEOUT_S (peer-to-peer transport)."""


class ks_params(C.Structure):
    _fields_ = [
        ("nev", C.c_int32), ("which", C.c_int32), ("tol", C.c_double), ("mindim", C.c_int32), ("maxdim", C.c_int32),
        ("restarts", C.c_int32), ("start_from", C.c_int32), ("initialize", C.c_int32), ("reserved", C.c_int32),
    ]


class ks_history(C.Structure):
    _fields_ = [
        ("mvproducts", C.c_int32), ("nconverged", C.c_int32), ("converged", C.c_int32), ("nev", C.c_int32),
        ("restarts", C.c_int32), ("reorth", C.c_int32), ("breakdowns", C.c_int32), ("explicit_steps", C.c_int32),
        ("seconds_expand", C.c_double), ("seconds_host", C.c_double), ("seconds_rotate", C.c_double),
    ]


class ks_expand_stats(C.Structure):
    _fields_ = [("steps", C.c_int32), ("reorth", C.c_int32), ("breakdowns", C.c_int32), ("explicit_steps", C.c_int32)]


HOST_APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
DEVICE_APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                               C.POINTER(C.c_void_p), C.POINTER(C.c_int64))

# name -> argtypes ; every function returns int except ks_last_error_string
vp, i32, i64, u64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
P = C.POINTER
PROTOTYPES = {
    "ks_version": [P(C.c_int), P(C.c_int)],
    "ks_ctx_create": [i32, P(vp)],
    "ks_comm_unique_id": [vp],
    "ks_ctx_create_dist": [i32, i32, i32, vp, P(vp)],
    "ks_ctx_create_hostcomm": [i32, i32, i32, HOST_ALLREDUCE_FN, HOST_EXCHANGE_FN, vp, P(vp)],
    "ks_ctx_create_p2p": [i32, i32, i32, P(vp)],
    "ks_ctx_p2p_handle": [vp, vp],
    "ks_ctx_p2p_attach": [vp, vp],
    "ks_ctx_destroy": [vp],
    "ks_ctx_synchronize": [vp],
    "ks_ctx_rank": [vp, P(C.c_int), P(C.c_int)],
    "ks_ctx_stream": [vp, P(vp)],
    "ks_operator_csr": [vp, i64, i64, i64, vp, vp, vp, i32, i32, i32, i32, P(vp)],
    "ks_operator_csr_dist": [vp, i64, i64, i64, vp, vp, vp, i32, i32, vp, vp, vp, vp, P(vp)],
    "ks_operator_dense": [vp, i64, vp, i64, i32, i32, P(vp)],
    "ks_operator_host_callback": [vp, i64, i32, HOST_APPLY_FN, vp, P(vp)],
    "ks_operator_device_callback": [vp, i64, i32, DEVICE_APPLY_FN, vp, P(vp)],
    "ks_operator_lu": [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, P(vp)],
    "ks_operator_lu_info": [vp, P(i64), P(i64), P(i64), P(i64)],
    "ks_operator_lu_layout": [vp, i32, P(i64), P(i64), P(i64), P(C.c_int)],
    "ks_operator_destroy": [vp],
    "ks_operator_size": [vp, P(i64), P(i64), P(C.c_int)],
    "ks_operator_format": [vp, P(C.c_double), P(C.c_int), P(C.c_int)],
    "ks_operator_apply_raw": [vp, vp, vp],
    "ks_workspace_create": [vp, i64, i64, i64, i32, i32, P(vp)],
    "ks_workspace_placement": [vp, P(C.c_int), P(C.c_double), P(C.c_double), P(C.c_int)],
    "ks_workspace_passes": [vp, P(C.c_int)],
    "ks_workspace_set_passes": [vp, i32, dbl],
    "ks_workspace_set_sstep": [vp, i32, dbl, dbl],
    "ks_debug_blk_time": [vp, i32, i32, i32, i32, i32, P(C.c_double), P(C.c_int)],
    "ks_workspace_sstep_info": [vp, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_double)],
    "ks_workspace_relation_info": [vp, P(C.c_int), P(C.c_double)],
    "ks_workspace_relation_probes": [vp, P(C.c_int)],
    "ks_workspace_fused_rotations": [vp, P(C.c_int), P(C.c_int), P(C.c_int)],
    "ks_workspace_split_rotations": [vp, P(C.c_int)],
    "ks_workspace_deflated_blocks": [vp, P(C.c_int), P(C.c_int)],
    "ks_sstep_partition": [C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_int), C.c_int, P(C.c_int)],
    "ks_workspace_assert_arnoldi": [vp, i32],
    "ks_workspace_provenance": [vp, P(C.c_int)],
    "ks_workspace_check_guard": [vp, P(C.c_int)],
    "ks_workspace_destroy": [vp],
    "ks_workspace_dims": [vp, P(i64), P(C.c_int), P(C.c_int), P(i64)],
    "ks_workspace_H": [vp, P(vp), P(C.c_int)],
    "ks_workspace_Q": [vp, P(vp), P(C.c_int)],
    "ks_workspace_col_ptr": [vp, i32, P(vp)],
    "ks_workspace_set_seed": [vp, u64],
    "ks_col_upload": [vp, i32, vp],
    "ks_col_download": [vp, i32, vp],
    "ks_cols_download": [vp, i32, i32, vp, i64],
    "ks_cols_upload": [vp, i32, i32, vp, i64],
    "ks_col_fill_uniform": [vp, i32, u64],
    "ks_col_norm": [vp, i32, P(dbl)],
    "ks_col_div": [vp, i32, dbl],
    "ks_col_copy": [vp, i32, i32],
    "ks_apply": [vp, vp, i32, i32],
    "ks_gemv_t": [vp, i32, i32, vp],
    "ks_gemv_n_sub": [vp, i32, i32, vp],
    "ks_rotate": [vp, i32, i32, i32, vp, i32],
    "ks_basis_times": [vp, i32, i32, vp, i32, i32, vp, i64],
    "ks_orthogonalize": [vp, i32, P(C.c_int)],
    "ks_reinitialize": [vp, i32, vp, P(C.c_int)],
    "ks_iterate_arnoldi": [vp, vp, i32, i32, P(ks_expand_stats)],
    "ks_params_default": [i64, P(ks_params)],
    "ks_partialschur": [vp, vp, P(ks_params), vp, vp, P(ks_history)],
    "ks_restart": [vp, P(ks_params), i32, P(C.c_int), P(C.c_int), P(C.c_int), vp, vp, vp],
    "ks_expand_restart": [vp, vp, P(ks_params), i32, i32, P(C.c_int), P(C.c_int), P(C.c_int), vp, vp, vp, P(ks_expand_stats), vp],
    "ks_profile_enable": [vp, i32],
    "ks_profile_reset": [vp],
    "ks_profile_get": [vp, i32, vp, vp, vp],
    "ks_residual_norms": [vp, vp, i32, P(dbl), P(dbl)],
    "ks_arnoldi_relation": [vp, vp, i32, P(dbl), P(dbl)],
    "ks_host_schurfact": [i32, vp, i32, i32, i32, i32, i32, vp, i32, i32],
    "ks_host_restart_step": [i32, vp, i32, vp, i32, i32, i32, i32, i32, dbl, i32, P(C.c_int), P(C.c_int), P(C.c_int), vp, vp, vp],
    "ks_host_sortschur": [i32, vp, i32, i32, i32, vp, i32, i32, i32, i32],
    "ks_host_givens": [i32, vp, vp, P(dbl), vp, vp],
    "ks_last_words": [C.c_char_p, i32],
}

_lib = None


def load():
    """dlopen libkschur_hip.so (built by __graft_entry__.build() / arnoldimethod.jl_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the MI355X HIP library has not been built (run `python __graft_entry__.py build`). "
            "There is no CPU fallback for the product path."
        )
    # default (RTLD_LOCAL) on purpose: with RTLD_GLOBAL the ROCm runtime's symbols interpose other extension
    # modules of the process and the interpreter aborted at exit ("double free or corruption") in a GPU-less run
    L = C.CDLL(LIB_PATH)
    L.ks_last_error_string.restype = C.c_char_p
    L.ks_last_error_string.argtypes = []
    for name, args in PROTOTYPES.items():
        fn = getattr(L, name)  # AttributeError here == header / library mismatch
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = L
    return L


def check(rc: int):
    if rc == KS_OK:
        return
    msg = load().ks_last_error_string().decode(errors="replace")
    if rc == KS_ERR_ARGUMENT:
        raise ArgumentError(msg)
    if rc == KS_ERR_DIMENSION:
        raise DimensionMismatch(msg)
    if rc == KS_ERR_QR:
        raise QRDidNotConverge(msg)
    if rc == KS_ERR_COMM:
        raise CommTimeout(msg)
    raise HipError(f"libkschur_hip error {rc}: {msg}")
