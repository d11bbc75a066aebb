"""`-m "not gpu"`: the C-ABI library loads and exports every symbol include/kschur.h declares, the
ctypes prototype table matches the header, and the product path fails LOUDLY without a GPU (no CPU
fallback, no oracle behind the API)."""
import ctypes as C
import os
import re

import pytest

from __graft_entry__ import ROOT, import_package

pkg = import_package()
_lib = pkg._lib


def header_functions():
    txt = open(os.path.join(ROOT, "include", "kschur.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int)\s+(ks_[A-Za-z0-9_]+)\s*\(", txt, flags=re.M)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run `python __graft_entry__.py build` first"
    L = C.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 45
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/kschur.h but not exported"


def test_prototype_table_matches_header():
    names = set(header_functions()) - {"ks_last_error_string"}
    assert names == set(_lib.PROTOTYPES), (names ^ set(_lib.PROTOTYPES))


def test_version_and_error_string():
    L = _lib.load()
    ma, mi = C.c_int(), C.c_int()
    assert L.ks_version(C.byref(ma), C.byref(mi)) == 0
    assert (ma.value, mi.value) == (0, 1)
    assert isinstance(L.ks_last_error_string(), bytes)


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_product_path_fails_loudly_without_gpu():
    import numpy as np
    import scipy.sparse as sp

    L = _lib.load()
    h = C.c_void_p()
    rc = L.ks_ctx_create(0, C.byref(h))
    assert rc in (_lib.KS_ERR_NO_DEVICE, _lib.KS_ERR_HIP)
    assert b"no CPU fallback" in L.ks_last_error_string() or rc == _lib.KS_ERR_HIP
    with pytest.raises(_lib.HipError):
        pkg.partialschur(sp.identity(20, format="csr") * 2.0, nev=2)
    # argument validation happens before any device work, with the reference's exception kinds
    with pytest.raises(pkg.DimensionMismatch):
        pkg.partialschur(np.zeros((4, 3)))
    with pytest.raises(pkg.ArgumentError):
        pkg.partialschur(np.zeros((6, 6)), nev=10)


def test_product_does_not_import_oracle():
    """The product package must not reference anything under oracle/ (parity claims depend on it)."""
    pk = os.path.join(ROOT, "arnoldimethod.jl_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".jl")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert '#include "../../oracle' not in src and "oracle/" not in src.replace("no oracle", ""), f
