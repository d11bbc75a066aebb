"""Builds libkschur_hip.so for gfx950 in-tree (explicit hipcc; no JIT cache).

Four translation units, compiled in parallel and linked into ONE shared library: the main unit (csrc/ks_hip.hip: C ABI, host
side, most kernels) and three parts of csrc/ks_block_inst.hip (the ~200 instantiations of the two streaming kernels of the
s-step expansion: Float64 block sizes 1-4, Float64 5 / 8 / 10, ComplexF64)."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "ks_hip.hip")
BLK = os.path.join(CSRC, "ks_block_inst.hip")
HEADERS = [os.path.join(CSRC, f) for f in ("ks_kernels.hpp", "ks_spmv_march.hpp", "ks_p2p.hpp", "ks_driver.hpp", "ks_smalldense.hpp", "ks_context.hpp", "ks_operators.hpp", "ks_sptrsv.hpp",
                                           "ks_workspace.hpp", "ks_backend.hpp", "ks_block.hpp", "ks_block_kernels.hpp", "ks_block_launch.hpp", "ks_block_mfma.hpp")]
DEPS = [SRC, BLK] + HEADERS + [os.path.join(HERE, "..", "include", "kschur.h")]
OUT = os.path.join(HERE, "libkschur_hip.so")
OBJDIR = os.path.join(HERE, "build")   # git-ignored and gpurun-ignored scratch: only the .so travels
# object -> (source, extra flags, the files it depends on)
BLK_DEPS = [BLK] + [os.path.join(CSRC, f) for f in ("ks_kernels.hpp", "ks_p2p.hpp", "ks_block_kernels.hpp", "ks_block_launch.hpp", "ks_block_mfma.hpp")]
UNITS = {
    "ks_hip.o": (SRC, [], DEPS),
    "ks_block_inst0.o": (BLK, ["-DKS_BLK_PART=0"], BLK_DEPS),
    "ks_block_inst1.o": (BLK, ["-DKS_BLK_PART=1"], BLK_DEPS),
    "ks_block_inst2.o": (BLK, ["-DKS_BLK_PART=2"], BLK_DEPS),
}


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


def _stale(obj: str, deps) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    extra = os.environ.get("KS_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DKS_FIN_TIMING (device-side stage timers of k_fin_step_t)
    jobs = []
    for name, (src, flags, deps) in UNITS.items():
        obj = os.path.join(OBJDIR, name)
        if force or extra or _stale(obj, deps):
            jobs.append([cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj] + flags + extra)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(OBJDIR, n) for n in UNITS] + ["-o", OUT, "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    run(link)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
