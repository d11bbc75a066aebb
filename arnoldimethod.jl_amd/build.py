"""Builds libkschur_hip.so for gfx950 in-tree (explicit hipcc; no JIT cache)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ks_hip.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("ks_kernels.hpp", "ks_p2p.hpp", "ks_driver.hpp", "ks_smalldense.hpp", "ks_context.hpp", "ks_operators.hpp", "ks_sptrsv.hpp", "ks_workspace.hpp", "ks_backend.hpp")] + [
    os.path.join(HERE, "..", "include", "kschur.h")
]
OUT = os.path.join(HERE, "libkschur_hip.so")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC, "-o", OUT, "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    cmd += os.environ.get("KS_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DKS_FIN_TIMING (device-side stage timers of k_fin_step_t)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=os.path.dirname(SRC))
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
