"""`-m "not gpu"`: the marching stencil SpMV (csrc/ks_spmv_march.hpp) issues its loads from inline assembly so that hipcc does not
wait for them; hipcc therefore does not KNOW the destination registers are in flight, and twice during development it placed a
register copy or a v_bfe of a loaded mask between the load and the hand-counted s_waitcnt (wrong rows on the GPU, nothing at
compile time).  This test compiles the kernel's instantiations for gfx950 to assembly and audits every one of them
(tools/isa_audit.py): between an inline-assembly global_load and the next s_waitcnt vmcnt, no compiler-scheduled instruction may
touch the loaded registers.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

from __graft_entry__ import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_audit  # noqa: E402

INSTANCES = [(7, 3), (5, 2), (3, 1), (1, -1), (4, -1), (7, -1), (8, -1)]
WINDOW_INSTANCES = [(7, 0x3E, 0x14, 3), (7, 0x3E, 0x36, 3), (7, 0x3E, 0x3E, 3)]   # the window form the solver launches for 3-D 7-point shapes
ZMARCH_INSTANCES = [(7, 0x3E, 0x14, 3, 0, 6), (7, 0x3E, 0x36, 3, 0, 6), (7, 0x3E, 0x3E, 3, 0, 6)]   # the z-marching form (large planes)
NEAR_INSTANCES = [(7, 3, 2, 4), (5, 2, 1, 3)]   # the +-1 taps from the neighbouring lanes (tools/spmv_slab.hip measures it; not dispatched)


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    d = tmp_path_factory.mktemp("isa")
    src = d / "march_inst.hip"
    inst = "\n".join(f"template __global__ void ksd::k_spmv_stencil_march<{ns}, {ko}>(const uint16_t*, const ksd::StencilDict<double>, const double*, double*, "
                     f"int64_t, int, const ksd::DevState*, int, double, double);" for ns, ko in INSTANCES)
    inst += "\n" + "\n".join(f"template __global__ void ksd::k_spmv_stencil_march<{a}, {b}, {c}, {e}>(const uint16_t*, const ksd::StencilDict<double>, const double*, double*, "
                             f"int64_t, int, const ksd::DevState*, int, double, double);" for a, b, c, e in NEAR_INSTANCES)
    inst += "\n" + "\n".join(f"template __global__ void ksd::k_spmv_stencil_marchz<{a}, {b}u, {c}u, {e}, {f}, {g}>(const uint16_t*, const ksd::StencilDict<double>, const double*, double*, "
                             f"int64_t, int, const ksd::DevState*, int, double, double);" for a, b, c, e, f, g in ZMARCH_INSTANCES)
    inst += "\n" + "\n".join(f"template __global__ void ksd::k_spmv_stencil_marchw<{a}, {b}u, {c}u, {e}>(const uint16_t*, const ksd::StencilDict<double>, const double*, double*, "
                             f"int64_t, int, const ksd::DevState*, int, double, double);" for a, b, c, e in WINDOW_INSTANCES)
    src.write_text(f'#include "{ROOT}/arnoldimethod.jl_amd/csrc/ks_spmv_march.hpp"\n{inst}\n')
    out = d / "march_inst.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", str(src), "-o", str(out)])
    return out.read_text()


@pytest.mark.parametrize("a,b,c,e", WINDOW_INSTANCES)
def test_the_window_form(listing, a, b, c, e):
    """register loads of the window form: the two far pairs and the masks, at three issue sites (the window pieces are LDS copies:
    no register in flight); and its LDS reads sit behind the hand-written wait + barrier (the audit treats any vmcnt wait as one)"""
    nloads, bad = isa_audit.audit(listing, f"k_spmv_stencil_marchwILi{a}ELj{b}ELj{c}ELi{e}E")
    assert nloads >= 3 * 3 and nloads % 3 == 0, nloads
    assert not bad, bad[:5]


@pytest.mark.parametrize("a,b,c,e,f,g", ZMARCH_INSTANCES)
def test_the_z_marching_form(listing, a, b, c, e, f, g):
    """register loads of the z-marching form: the own pair of the plane below (once, in the prologue) and the masks (prologue + the
    two halves of the unrolled loop); windows are LDS copies"""
    nloads, bad = isa_audit.audit(listing, f"k_spmv_stencil_marchzILi{a}ELj{b}ELj{c}ELi{e}ELi{f}ELi{g}E")
    assert nloads >= 4, nloads
    assert not bad, bad[:5]


@pytest.mark.parametrize("a,b,c,e", NEAR_INSTANCES)
def test_the_form_with_neighbour_lane_taps(listing, a, b, c, e):
    nloads, bad = isa_audit.audit(listing, f"k_spmv_stencil_marchILi{a}ELi{b}ELi{c}ELi{e}E")
    nl = a - 2 + 1 + 1   # two pairs less, the edge load, the masks
    assert nloads >= 3 * nl and nloads % nl == 0, nloads
    assert not bad, bad[:5]


@pytest.mark.parametrize("ns,ko", INSTANCES)
def test_no_compiler_instruction_touches_a_register_in_flight(listing, ns, ko):
    tag = f"k_spmv_stencil_marchILi{ns}ELi{ko}ELin1ELin1E" if ko >= 0 else f"k_spmv_stencil_marchILi{ns}ELin{-ko}ELin1ELin1E"
    nloads, bad = isa_audit.audit(listing, tag)
    # at least three issue sites (prologue, the two halves of the unrolled loop; hipcc duplicates them for the short kernels), each:
    # one load per slot (+ the own pair) + the masks
    nl = ns + (1 if ko < 0 else 0) + 1
    assert nloads >= 3 * nl and nloads % nl == 0, nloads
    assert not bad, bad[:5]


def test_the_audit_catches_a_touch():
    text = "\n".join(["_Z4kernv:", "\t;;#ASMSTART", "\tglobal_load_dwordx4 v[4:7], v1, s[2:3]", "\t;;#ASMEND", "\tv_mov_b64_e32 v[8:9], v[4:5]",
                      "\t;;#ASMSTART", "\ts_waitcnt vmcnt(0)", "\t;;#ASMEND", "\tv_mul_f64 v[8:9], v[4:5], v[6:7]", ".Lfunc_end0:"])
    nloads, bad = isa_audit.audit(text, "kern")
    assert nloads == 1 and len(bad) == 2 and bad[0][2] == 4
