"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` : oracle vs the reference's known-answer tests, host logic, C-ABI symbols.
`-m gpu`       : parity tests proper -- the HIP path (through the C-ABI) vs the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    """The suites need the in-tree shared libraries (git-ignored build artefacts).  Build them on demand so
    `pytest` works on a fresh clone: hipcc cross-compiles for gfx950 without a GPU; the oracle's C++ checker
    needs only g++."""
    import importlib.util

    lib = os.path.join(ROOT, "arnoldimethod.jl_amd", "libkschur_hip.so")
    spec = importlib.util.spec_from_file_location("_ks_build", os.path.join(ROOT, "arnoldimethod.jl_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    try:
        if b.needs_build():
            b.build()
    except Exception as e:  # noqa: BLE001 - e.g. no hipcc on this box: use what travelled with the snapshot
        if not os.path.exists(lib):
            raise RuntimeError(f"libkschur_hip.so is missing and could not be built: {e}")
    ref = os.path.join(ROOT, "oracle", "_build", "libkschur_cpuref.so")
    if not os.path.exists(ref):
        from oracle import cpuref

        cpuref.build()


def pytest_configure(config):
    _ensure_built()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
