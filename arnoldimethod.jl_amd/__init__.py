"""arnoldimethod.jl_amd -- MI355X-native Krylov-Schur Arnoldi hot path behind ArnoldiMethod.jl's API.

The directory name contains a dot, so import it through `__graft_entry__.import_package()` (which
registers it as `arnoldimethod_jl_amd`).  Importing the package does not touch the GPU; the first
API call dlopens `libkschur_hip.so` and fails loudly if it (or a gfx950 device) is missing.
"""
from ._lib import ArgumentError, CommTimeout, DimensionMismatch, HipError, QRDidNotConverge  # noqa: F401
from .api import (  # noqa: F401
    LI, LM, LR, SI, SR, ArnoldiWorkspace, Context, History, Operator, PartialSchur, Target, as_operator,
    csr_operator, default_context, dense_operator, device_operator, host_operator, lu_operator, splu_operator, partialeigen, partialschur, partialschur_, sstep_partition, vtype,
)
from . import matrices  # noqa: F401
# `extras` (ready-made device operators on the callback seam; needs torch + rocSPARSE) is imported on demand:
#     from arnoldimethod_jl_amd import extras

__all__ = [
    "partialschur", "partialschur_", "partialeigen", "ArnoldiWorkspace", "PartialSchur", "History",
    "LM", "LR", "SR", "LI", "SI", "Context", "Operator", "csr_operator", "dense_operator", "host_operator", "device_operator", "lu_operator", "splu_operator", "as_operator",
    "ArgumentError", "DimensionMismatch", "CommTimeout", "matrices", "sstep_partition",
]
