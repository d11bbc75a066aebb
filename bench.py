#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configuration.

    metric   : Arnoldi iterations / second (one iteration = one operator apply + orthogonalize!,
               src/expansion.jl:119-130), steady state over full Krylov-Schur restart cycles
               INCLUDING the host Schur/reorder work and the restart rotation
    workload : 3-D 7-point Laplacian, 216^3 = 10 077 696 rows (the "n = 10^7" headline case),
               nev = 20, which = :SR, mindim = 20, maxdim = 40, Float64, explicit start vector
    step     : one restart cycle = expand the Krylov basis from k+1 to maxdim, restart to k

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU; rows of A and V are block-partitioned
     over the ranks -- total work fixed => "scaling": "strong", as north_star asks.  The per-step
     exchanges have two transports, RCCL and the library's peer-to-peer regions; both are measured
     with the full W + K protocol -- RCCL FIRST and alone, each pass under a watchdog -- and reported
     under "transports", `value` is the faster valid one.  With 8 ranks (or --config5) the line also
     carries BASELINE config 5, the 464^3 Laplacian row-partitioned over the ranks, as "config5".)

The line validates itself: after the timed cycles the reference's two invariants (test/expansion.jl:29-30) are
evaluated ON THE BENCHED WORKSPACE (`validation`), and a run whose state violates them exits non-zero.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel class (algorithmic bytes per
launch / HIP-event duration on the library's own stream) and, as `fused_step`, the north-star quantity
(bytes of the fused SpMV + DGKS step / expansion wall time).  `cpu_baseline` times the reference's
un-fused op sequence on the host cores (oracle/cpu_backend.cpp, kind "port") on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import threading
import time



def _cpu_budget():
    """CPUs this process may actually USE: the cgroup CFS quota when there is one (a container that sees 256 logical
    CPUs but is limited to 16 CPUs' worth of time runs a 128-thread OpenMP team SLOWER than a 16-thread one: the
    spinning threads burn the quota -- measured on the GPU box: 5.3 vs 35 iterations/s for the CPU baseline)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


# the CPU baseline's OpenMP team (and OpenBLAS behind scipy's ARPACK): as many threads as the container may run, one per
# core, pinned -- all of this must be in the environment before libgomp / OpenBLAS are loaded
CPU_BUDGET = _cpu_budget()
if CPU_BUDGET < (os.cpu_count() or 1):
    os.environ.setdefault("OMP_NUM_THREADS", str(CPU_BUDGET))
    os.environ.setdefault("OPENBLAS_NUM_THREADS", str(CPU_BUDGET))
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling


KERNEL_NAMES = {
    "dots": "k_dots (h = V'w, |w|^2)",
    "axpy": "k_axpy (w -= V c, |w|^2; second DGKS pass)",
    "fused": "k_axpy_dots_cs (w -= V h, |w|^2 and c = V'w of the second DGKS pass, V read once)",
    "spmv": "SpMV kernel of the layout in use (config.spmv_layout; stencil-mask, Float64, one GPU: k_spmv_stencil_march*)",
    "scale": "k_scale",
    "rotate": "k_rotate_fma (restart rotation V <- V Q in place, src/run.jl:363-365)",
}


BLOCK_KERNEL_NAMES = {
    "dots": "k_bdots_mfma (s-step pass 1: P = S'Z and Z'Z for a block of s vectors, basis read once per block)",
    "rotate": "k_brotdots_mfma (restart rotation V <- V (T Q), src/run.jl:363-365, fused with s-step pass 1 of the next block: old basis and block read once, rotated columns written, P = S'Z and Z'Z accumulated in the same sweep)",
    "fused": "k_bupdate_mfma (s-step pass 2: block = (Z - S coef) R1^-1 written in place, C = S'block and its Gram matrix; basis read once per block)",
}


def pmc_traffic(kernel_class):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE collected separately, corrected as MI355X_MICROARCH.md prescribes; tools/pmc_summary.py)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        e = d["classes"][kernel_class]
        return {"bytes_per_launch": e["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"],
                "ratio": e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], "source": "profiles/pmc_traffic.json (" + d.get("source", "") + ")"}
    except Exception:
        return None


def step_bytes(n, nnz, j, reorth, bpn=12.0):
    """Algorithmic bytes of one Arnoldi step at basis size j (SURVEY.md 8d / BASELINE.md section 4);
    `bpn` = bytes the SpMV streams per stored non-zero in the layout the library chose (12 plain CSR,
    4 value-indexed)."""
    b = bpn * nnz + 4.0 * (n + 1) + 8.0 * n * (2 * j + 2)
    if reorth:
        b += 16.0 * j * n + 16.0 * n
    return b


def spmv_bytes(fmt, nnz, n):
    """Bytes ONE product y = A x streams in the layout the library chose -- the same figure the library books per launch
    (roofline.spmv.bytes_per_launch): matrix bytes + row pointers WHERE THE LAYOUT HAS THEM + x and y.  The stencil-mask layout
    keeps one mask byte per row and no row pointers (17 B per row for the 7-point Laplacian); sliced ELLPACK one pointer per
    64-row slice."""
    lay = fmt["layout"]
    ptr = 0.0 if lay == "stencil" else (4.0 * (n / 64.0 + 1) if lay.startswith("sell") else 4.0 * (n + 1))
    return fmt["bytes_per_nnz"] * nnz + ptr + 16.0 * n


def plain_csr_spmv(pkg, ctx, A_host, n, reps=20):
    """y = A x with the matrix uploaded as PLAIN CSR (rowptr / colidx int32 / val f64: 12 B per non-zero, SURVEY 8d's
    formula; the row-block kernel k_spmv_csr: coalesced index/value reads, products through LDS, per-row sums), `reps`
    launches on the library's stream between HIP events.  Returns GB/s on its algorithmic bytes 12 nnz + 4 (n+1) + 16 n."""
    import torch

    from arnoldimethod_jl_amd import _lib

    old = os.environ.get("KS_SPMV_FORMAT")
    os.environ["KS_SPMV_FORMAT"] = "csr"
    try:
        op = pkg.csr_operator(pkg.matrices.to_scipy(*A_host, n), ctx)
    finally:
        if old is None:
            os.environ.pop("KS_SPMV_FORMAT", None)
        else:
            os.environ["KS_SPMV_FORMAT"] = old
    try:
        fmt = op.format
        x = torch.from_numpy(pkg.matrices.start_vector(n)).cuda()
        y = torch.empty_like(x)
        torch.cuda.synchronize()
        L = _lib.load()
        for _ in range(3):
            _lib.check(L.ks_operator_apply_raw(op._h, x.data_ptr(), y.data_ptr()))
        ctx.synchronize()
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(reps):
            _lib.check(L.ks_operator_apply_raw(op._h, x.data_ptr(), y.data_ptr()))
        p = ctx.profile_get()["spmv"]
        ctx.profile_enable(False)
        ctx.profile_reset()
        ms = p["ms"] / max(p["count"], 1)
        gbs = p["bytes"] / max(p["count"], 1) / (ms * 1e-3) / 1e9
        return {"layout": fmt["layout"], "bytes_per_nnz": fmt["bytes_per_nnz"], "launches": p["count"], "avg_launch_ms": ms,
                "bytes_per_launch": p["bytes"] / max(p["count"], 1), "GBps": gbs, "frac": gbs / HBM_PEAK_GBS, "measured_in_run": True}
    finally:
        op.close()


def shift_invert_record(pkg, nx=200, ny=250, reps=20):
    """BASELINE config 4's operator in its general form, measured in the run (rank 0, one GPU): a 2-D Laplacian + i*diag,
    sigma interior, ComplexF64; SuperLU factorisation on the host (as the reference's users factor on the host), the two
    sparse triangular solves of every product on the device (ks_operator_lu).  Reports ms per product next to the host solve
    of the same factorisation and the difference between the two results.  A side record: never part of `value`."""
    import time

    import numpy as np
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    n = nx * ny
    rng = np.random.default_rng(3)
    ex = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx))
    ey = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(ny, ny))
    A = (sp.kron(sp.identity(ny), ex) + sp.kron(ey, sp.identity(nx))).astype(np.complex128) + 1j * sp.diags(0.3 * rng.random(n))
    sigma = 1.7 + 0.1j
    M = (A - sigma * sp.identity(n)).tocsc()
    t = time.perf_counter()
    lu = spla.splu(M, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    t_factor = time.perf_counter() - t
    ctx = pkg.Context(0)
    op = pkg.splu_operator(lu, ctx)
    try:
        ws = pkg.ArnoldiWorkspace(n, 4, np.complex128, ctx=ctx)
        b = rng.random(n) + 1j * rng.random(n)
        ws.set_col(0, b)
        ws.apply(op, 0, 1)
        y = ws.col(1)
        t = time.perf_counter()
        x = lu.solve(b)
        t_host = time.perf_counter() - t
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            ws.apply(op, 0, 1)
        ctx.synchronize()
        t_dev = (time.perf_counter() - t) / reps
        info = op.lu_info
        return {"workload": f"(A - sigma I)^-1 x, A = laplace2d {nx}x{ny} + i*diag, sigma = 1.7+0.1i, ComplexF64, SuperLU factors applied by ks_operator_lu",
                "n": n, "stored_entries": info["nnz_l"] + info["nnz_u"], "dependency_levels": info["levels_l"] + info["levels_u"],
                "ms_per_product": 1e3 * t_dev, "host_solve_ms": 1e3 * t_host, "host_factorisation_s": t_factor, "products_timed": reps,
                "max_rel_diff_vs_host_solve": float(np.abs(y - x).max() / np.abs(x).max()),
                "repeatable": bool(np.array_equal(ws.col(1), y)), "measured_in_run": True}
    finally:
        op.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=216, help="grid points per dimension (216^3 ~ 1e7 rows)")
    ap.add_argument("--nev", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shift-invert", action="store_true", help="skip the side record of the sparse shift-invert operator (N = 1 only)")
    ap.add_argument("--no-profile", action="store_true", help="skip the second (HIP-event instrumented) pass")
    ap.add_argument("--sync-cycles", action="store_true",
                    help="synchronise the device after every timed restart cycle (rounds 1-6a).  Default off: the library's own "
                         "driver (ks_partialschur) runs its cycles back to back -- the speculative chain of the next expansion is still on "
                         "the device when the restart returns -- and the timed region is bracketed by a barrier + synchronize on both sides "
                         "(any number of ranks)")
    ap.add_argument("--sstep", type=int, default=int(os.environ.get("KS_BENCH_SSTEP", "20")),
                    help="s-step (block) expansion: steps per block (ks_workspace_set_sstep; 0 = the per-step expansion of rounds 2-3)")
    ap.add_argument("--config5", action="store_true", help="also measure BASELINE config 5 (464^3 over the ranks) as a second record; "
                                                           "default: only with 8 ranks")
    ap.add_argument("--config5-grid", type=int, default=464)
    args = ap.parse_args()

    import torch

    from __graft_entry__ import import_package

    pkg = import_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # KS_SAME_DEVICE=1 (testing on a one-GPU box): all ranks share device 0; RCCL refuses that, so the
    # rendezvous runs over gloo and only the peer-to-peer transport can be measured
    same_device = os.environ.get("KS_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if same_device else "cuda"
    dist = None
    force_dist = os.environ.get("KS_FORCE_DIST") == "1"  # run the sharded code path on a single rank (debugging)
    if world > 1:
        # last line of defence for N > 1: whatever hangs OUTSIDE the per-pass watchdogs (the rendezvous itself, a vote, the
        # teardown), rank 0 still prints a line -- value null, the reason -- and every rank leaves
        def _global_watchdog():
            limit = float(os.environ.get("KS_BENCH_GLOBAL_DEADLINE_S", "1500"))
            time.sleep(limit)
            if rank == 0:
                print(json.dumps({"metric": "arnoldi_iters_per_sec", "value": None, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
                                  "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                                  "dtype": "f64", "data": "synthetic", "config": {"workload": f"laplace3d-7pt {args.grid}^3"},
                                  "error": f"bench.py did not finish within {limit:.0f} s (hung rendezvous or teardown?)"}), flush=True)
            os._exit(3)

        threading.Thread(target=_global_watchdog, daemon=True).start()
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # CONTROL PLANE over gloo (rendezvous, the unique id / IPC handles, the MAX over ranks of the timing, pass/fail
        # votes: a few CPU scalars); the DATA PATH's collectives are the library's own RCCL communicator
        # (ks_ctx_create_dist: ncclAllReduce / ncclSend / ncclRecv) or its peer-to-peer regions.  One RCCL communicator
        # in the process instead of two: torch's would be the first thing to meet a new fabric, outside every watchdog.
        if os.path.isdir("/sys/class/net/lo"):  # one node: rendezvous over loopback (the container's hostname may not resolve)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if world > 1 and same_device:
            dist.init_process_group("gloo")
            # RCCL cannot run here: measure the peer-to-peer transport and the host-staged one (which executes the RCCL
            # transport's launch structure with the exchanges staged through gloo)
            os.environ["KS_BENCH_TRANSPORTS"] = ",".join(
                t for t in os.environ.get("KS_BENCH_TRANSPORTS", "host,p2p").split(",") if t in ("p2p", "host")) or "p2p"
        elif world > 1:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("gloo", rank=0, world_size=1)
        red_dev = "cpu"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    m = args.grid
    n = m ** 3
    nev, mindim, maxdim = args.nev, max(10, args.nev), max(20, 2 * args.nev)
    which = "SR"
    tol = float(np.sqrt(np.finfo(np.float64).eps))

    def measure(transport, m=m, n=n, steps=None, warmup=None, profile=True):
        """Build operand + workspace (rows block-partitioned over the ranks), run W untimed and K timed
        restart cycles, then the same K cycles once more with per-kernel HIP events."""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        if dist is None:
            ctx = pkg.Context(local_rank)
            ip, ix, dv = pkg.matrices.laplace3d_csr(m, m, m)
            nnz_global = int(ip[-1])
            A_host = (ip, ix, dv)
            op = pkg.csr_operator(pkg.matrices.to_scipy(ip, ix, dv, n), ctx)
            ws = pkg.ArnoldiWorkspace(n, maxdim, np.float64, ctx=ctx)
            v1 = pkg.matrices.start_vector(n)
        else:
            from arnoldimethod_jl_amd import dist as ksdist  # registered by import_package()

            ctx, op, ws, v1, nnz_global = ksdist.setup_laplace3d(pkg, dist, m, maxdim, local_rank, transport)
            A_host = None
        fmt = op.format
        placement = ws.placement
        basis_passes = ws.passes
        # s-step (block) expansion (include/kschur.h: ks_workspace_set_sstep): device-resident operator; with several ranks
        # the two reductions of a block are the only collectives of its s steps
        sstep = args.sstep if args.sstep >= 2 else 0
        ws.set_sstep(sstep)     # (0 included: the library's own default is ON)
        ws.reinitialize(0, v1)
        ws.iterate_arnoldi(op, 1, mindim)  # initial expansion, src/run.jl:267 (untimed)

        state = dict(k=mindim, active=0, steps=0, bytes=0.0, moved=0.0, t_expand=0.0, t_restart=0.0, reorth=0, trail=[], ritz=None)

        split_cycle = os.environ.get("KS_BENCH_SPLIT_CYCLE", "0") == "1"

        # (several ranks too: RCCL and the peer-to-peer transport speculate chains like one rank does, every rank enqueues the same
        # sequence; the host-staged transport drains its stream in every exchange anyway)
        sync_cycles = args.sync_cycles

        def cycle(timed, sync=True):
            k = state["k"]
            # one cycle of _partialschur's loop (src/run.jl:272-365) the way ks_partialschur runs it: expansion + restart
            # in one library call (with the explicit second pass, KS_PASSES=3, the restart's Schur factorisation overlaps the
            # tail of the expansion; with the default two-pass expansion H is final only when the batch ends)
            # (KS_BENCH_SPLIT_CYCLE=1: the two calls ks_iterate_arnoldi + ks_restart of rounds 1-2, bit-identical results)
            info0 = ws.sstep_info if (sstep and timed) else {}
            blocks0, fused0, chains0 = info0.get("blocks", 0), info0.get("fused_rotations", 0), info0.get("chains_adopted", 0)
            t0 = time.perf_counter()
            if split_cycle:
                st = ws.iterate_arnoldi(op, k + 1, maxdim)
                t1 = time.perf_counter()
                r = ws.restart(state["active"], nev, which, tol, mindim, maxdim)
            else:
                r = st = ws.expand_restart(op, k, state["active"], nev, which, tol, mindim, maxdim)
                t1 = t0 + r["seconds"][0]
            if sync:
                ctx.synchronize()
            t2 = time.perf_counter()
            if timed:
                nst = maxdim - k
                state["steps"] += nst
                state["reorth"] += st["reorth"]
                # all steps of this workload take the DGKS second pass; attribute per step when they do
                all_re = st["reorth"] == nst
                for j in range(k + 1, maxdim + 1):
                    state["bytes"] += step_bytes(n, nnz_global, j, all_re, fmt["bytes_per_nnz"])
                if not all_re:
                    jm = (k + 1 + maxdim) / 2.0
                    state["bytes"] += st["reorth"] * (16.0 * jm * n + 16.0 * n)
                # bytes the launched kernels MUST move -- the traffic-true figure.  Implicit second pass (default): TWO
                # passes over V per step (k_dots, k_axpy_dots_cs) whether or not the DGKS test asks for the second
                # projection = SURVEY 8d's compulsory B_step(j); KS_PASSES=3: a third one (k_axpy) when it does
                spmv_b = spmv_bytes(fmt, nnz_global, n)
                blk = []
                if sstep:
                    info = ws.sstep_info
                    blk = pkg.sstep_partition(np.float64, k + 1, nst, sstep)
                    if info["blocks"] - blocks0 != len(blk):   # a block was abandoned (or no shifts yet): not a block cycle
                        state["blk_irregular"] = state.get("blk_irregular", 0) + 1
                        blk = []
                    state["blk_cycles"] = state.get("blk_cycles", 0) + (1 if blk else 0)
                    state["blk_blocks"] = state.get("blk_blocks", 0) + len(blk)
                    state["blk_pivot_min"] = min(state.get("blk_pivot_min", 1.0), info["pivot_stage1"], info["pivot_stage2"])
                    state["blk_gram_dev"] = max(state.get("blk_gram_dev", 0.0), info["gram_dev"])
                    state["blk_s_now"], state["blk_abandoned"] = info["s"], info["abandoned"]
                if blk:
                    # s-step cycle: per block of s steps on kk columns  s products (the Newton shift is fused into the stencil
                    # kernel; other layouts pay a 24 n-byte pass per product) + k_bdots 8 n (kk + s) + k_bupdate 8 n (kk + 2 s)
                    shift_b = 0.0 if fmt["layout"] == "stencil" else 24.0 * n
                    kk = k + 1
                    # the rotation of the PREVIOUS restart ran fused with this cycle's first pass (k_brotdots_mfma): that launch
                    # reads the maxdim + 1 old columns and the block, writes the rotated ones (all but the locked prefix) --
                    # instead of pass 1's 8 n (kk + s); its bytes belong to this expansion's wall time
                    fused_rot = info["fused_rotations"] - fused0 > 0
                    state["fused_rotations"] = state.get("fused_rotations", 0) + (1 if fused_rot else 0)
                    # the first products of this expansion ran SPECULATIVELY during the previous cycle's host step (ks_backend.hpp:
                    # min(10, steps - 1) products enqueued behind the previous expansion): their bytes are part of `moved`, but they
                    # were executed inside a restart interval, not inside this expansion's wall time
                    if info.get("chains_adopted", 0) - chains0 > 0:
                        ne = spec_chain_products(nst)
                        state["spec_products"] = state.get("spec_products", 0) + ne
                        state["spec_bytes"] = state.get("spec_bytes", 0.0) + ne * (spmv_b + shift_b)
                    for ib, sb in enumerate(blk):
                        p1 = 8.0 * n * ((maxdim + 1) + sb + (kk - min(state["active"], kk - 1))) if (fused_rot and ib == 0) else 8.0 * n * (kk + sb)
                        state["moved"] += sb * (spmv_b + shift_b) + p1 + 8.0 * n * (kk + 2 * sb)
                        # FP64 work of the two block kernels per row: pass 1 (kk + (sb + 1) / 2) sb multiply-adds, pass 2
                        # (kk + sb) sb (the update) + kk sb (inner products) + sb (sb + 1) / 2 (Gram triangle)
                        state["blk_flops_dots"] = state.get("blk_flops_dots", 0.0) + 2.0 * n * (kk * sb + sb * (sb + 1) / 2.0)
                        state["blk_flops_fused"] = state.get("blk_flops_fused", 0.0) + 2.0 * n * ((kk + sb) * sb + kk * sb + sb * (sb + 1) / 2.0)
                        kk += sb
                for j in range(k + 1, maxdim + 1):
                    if not blk:
                        state["moved"] += spmv_b + 8.0 * n * (j + 1) + 8.0 * n * (j + 2)
                    # SURVEY 8d's compulsory B_step(j) as written there: plain CSR (12 B / non-zero), V twice, column once
                    state["survey"] = state.get("survey", 0.0) + 12.0 * nnz_global + 4.0 * (n + 1) + 8.0 * n * (2 * j + 2)
                if basis_passes == 3 and not blk:
                    state["moved"] += st["reorth"] * 8.0 * n * ((k + 1 + maxdim) / 2.0 + 2)
                state["t_expand"] += t1 - t0
                state["t_restart"] += t2 - t1
                state.setdefault("cycle_s", []).append(t2 - t0)
                state["trail"].append((r["k"], r["nlock"]))
                state["ritz"] = np.sort_complex(r["eigenvalues"][: r["k"]])
            state["k"], state["active"] = r["k"], r["nlock"]

        for _ in range(warmup):
            cycle(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            cycle(True, sync=sync_cycles)
        barrier()
        elapsed = time.perf_counter() - t0
        state["sync_cycles"] = bool(sync_cycles)
        if not sync_cycles:
            # the tail the closing barrier waited for (the last cycle's speculative chain) and the loop's own bookkeeping belong to the
            # timed region: booked to the restart intervals, so that t_expand + t_restart == elapsed and the whole-cycle fractions
            # (traffic_fractions: cycle_*) divide by the timed region exactly
            state["t_restart"] += max(0.0, elapsed - (state["t_expand"] + state["t_restart"]))
        # SELF-VALIDATION on the benched state (outside the timed region): the reference's two invariants of an Arnoldi /
        # Krylov-Schur decomposition, test/expansion.jl:29-30, evaluated on the device for the k columns the last restart
        # left -- ||A V_k - V_{k+1} H_k||_F <= 1e-11 ||H||_F (tol-level instead once vectors are locked: the locked part of
        # the relation holds to tol |lambda|, src/run.jl:206-208,360) and ||V'V - I||_F <= sqrt(eps) / 100
        rel, orth = ws.arnoldi_relation(op, state["k"])
        hnorm = float(np.linalg.norm(ws.H[: state["k"] + 1, : state["k"]]))
        lim_rel = (1e-11 if state["active"] == 0 else 10.0 * tol) * hnorm
        lim_orth = float(np.sqrt(np.finfo(np.float64).eps)) / 100.0
        validation = {"arnoldi_rel": rel / hnorm, "orth": orth, "k": state["k"], "locked": state["active"], "H_fro": hnorm,
                      "limit_rel": lim_rel / hnorm, "limit_orth": lim_orth, "ok": bool(rel <= lim_rel and orth <= lim_orth),
                      "what": "||A V_k - V_{k+1} H_k||_F / ||H_k||_F and ||V'V - I||_F of the benched workspace after the timed "
                              "cycles, evaluated on the device (test/expansion.jl:29-30)"}
        # Per-kernel HIP-event timing: the event pairs (recorded on the library's own stream around every
        # launch) cost ~4 % of throughput on this launch-dense path, so they are NOT left on while `value`
        # is measured; the same K cycles are repeated immediately afterwards with events on and the
        # per-kernel figures of `roofline` come from that second pass (same workload, same state machine).
        prof = None
        if profile and not args.no_profile:
            ctx.profile_reset()
            ctx.profile_enable(True)
            for _ in range(steps):
                cycle(False)
            prof = ctx.profile_get()
            ctx.profile_enable(False)
        # the SpMV BASELINE.json's metric names (north_star: "CSR SpMV with coalesced row-pointer/column reads and LDS
        # per-row partial sums"), measured in this run next to the layout the solver picked: the same matrix uploaded
        # as plain CSR (12 B per non-zero), 20 launches on the library's stream between HIP events
        spmv_csr = None
        if dist is None and profile and not args.no_profile:
            spmv_csr = plain_csr_spmv(pkg, ctx, A_host, n)
        if dist is not None and world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ws.close()
        op.close()
        ctx.close()
        return dict(elapsed=elapsed, state=state, prof=prof, nnz_global=nnz_global, A_host=A_host, fmt=fmt, placement=placement, basis_passes=basis_passes, sstep=sstep,
                    validation=validation, spmv_csr=spmv_csr, steps=steps, warmup=warmup)

    # N > 1: the row-partitioned solver has two transports for its per-step exchanges -- RCCL collectives
    # and the library's own peer-to-peer regions over xGMI (csrc/ks_p2p.hpp).  Both are measured with the
    # full W + K protocol, back to back; `value` is the faster pass that completed on every rank; when both
    # completed they must agree (same (k, nlock) sequence of the restarts, same Ritz values) or the slower-to-
    # verify one (p2p) is dropped.  A failure of EITHER transport -- an exception on any rank, or a pass that does
    # not finish within its deadline (a hung collective) -- is reported under "transports" and the line is printed
    # from the survivor; if nothing survives the line still appears, with value null and exit status 3.
    # KS_BENCH_TRANSPORTS=rccl restricts the run; KS_BENCH_INJECT_FAIL=rccl|p2p[:hang] makes that pass fail or
    # hang on purpose (tests/test_bench_line.py).
    passes = {}
    if dist is None or (world == 1 and "KS_BENCH_TRANSPORTS" not in os.environ):
        order = ["single"]
    else:  # (KS_FORCE_DIST=1 KS_BENCH_TRANSPORTS=rccl,p2p exercises this selection logic on a single rank)
        # RCCL FIRST AND ALONE (the transport every fabric supports; nothing of the peer-to-peer machinery -- IPC handles,
        # uncached regions, in-kernel spins -- has touched the devices when it runs), then the peer-to-peer regions
        order = [t for t in os.environ.get("KS_BENCH_TRANSPORTS", "rccl,p2p").split(",") if t in ("rccl", "p2p", "host")] or ["rccl"]
    inject = os.environ.get("KS_BENCH_INJECT_FAIL", "")
    deadline_s = float(os.environ.get("KS_BENCH_PASS_DEADLINE_S", "300"))

    def emit(out):
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        if rank == 0:
            print(json.dumps(out), flush=True)

    extra = {}

    def build_line(with_cpu_baseline=True):  # (the records gathered so far ride along)
        line = make_line(args, pkg, passes, order, world, rank, force_dist, dict(m=m, n=n, nev=nev, mindim=mindim, maxdim=maxdim, which=which),
                         with_cpu_baseline)
        line.update(extra)
        return line

    def run_pass(tr, store, key, fn):
        """One measured pass under a watchdog; the outcome (result dict or {"error": ...}) lands in store[key]."""
        ok, res, err = 1, None, ""
        done = threading.Event()

        def watchdog():
            # a pass that hangs inside a collective cannot be interrupted from Python: report what we have and leave
            if done.wait(deadline_s):
                return
            store[key] = {"error": f"pass did not finish within {deadline_s:.0f} s (hung exchange?)"}
            line = build_line(with_cpu_baseline=False)
            try:
                emit(line)
            finally:
                os._exit(0 if line["value"] is not None else 3)

        if tr != "single":
            threading.Thread(target=watchdog, daemon=True).start()
        try:
            if inject.split(":")[0] == tr:
                if inject.endswith(":crash"):  # the LAST rank dies the hard way (abort, as after a GPU memory fault); the others wait
                    if rank == world - 1:
                        os.abort()
                    time.sleep(10 * deadline_s)
                if inject.endswith(":hang"):
                    time.sleep(10 * deadline_s)
                raise RuntimeError(f"injected failure of the {tr} pass (KS_BENCH_INJECT_FAIL)")
            res = fn()
        except Exception as e:  # noqa: BLE001
            if tr == "single":
                raise
            ok, err = 0, f"{type(e).__name__}: {e}"
        if dist is not None and world > 1:
            try:
                t = torch.tensor([ok], dtype=torch.int32, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                if int(t.item()) == 0:
                    ok, err = 0, err or "failed on another rank"
            except Exception as e:  # noqa: BLE001 - the process group itself is broken
                ok, err = 0, err or f"{type(e).__name__}: {e}"
        store[key] = res if ok else {"error": err}
        done.set()

    def leave_last_words(next_pass):
        """N > 1, rank 0: should the process die under the next pass (a memory fault reported by the GPU runtime aborts it; the
        launcher sends SIGTERM when a peer died), the line built from what HAS completed is what the library writes to
        stdout on the way out (ks_last_words: write(2) + _exit from the signal handler)."""
        if world <= 1 or rank != 0:
            return
        try:
            line = build_line(with_cpu_baseline=False)
            line["died_during"] = next_pass
            lib = pkg._lib.load()
            lib.ks_last_words(json.dumps(line).encode(), 0 if line["value"] is not None else 3)
        except Exception:  # noqa: BLE001 - never let the safety net take the run down
            pass

    for tr in order:
        leave_last_words(tr)
        run_pass(tr, passes, tr, lambda tr=tr: measure(None if tr == "single" else tr))

    # BASELINE config 5: the 464^3 Laplacian (n = 99 897 344, V = 32.8 GB in total) row-partitioned over the ranks, nev 20,
    # 20/40 -- a second record in the same line, on the transport that won above, 1 untimed + 3 timed cycles
    if args.config5 or world == 8 or os.environ.get("KS_BENCH_CONFIG5") == "1":
        valid = [t for t in order if t in passes and "error" not in passes[t]]
        if valid:
            tr5 = min(valid, key=lambda t: passes[t]["elapsed"])
            g5 = args.config5_grid
            box = {}
            leave_last_words("config5")
            run_pass(tr5, box, "r", lambda: measure(None if tr5 == "single" else tr5, m=g5, n=g5 ** 3, steps=3, warmup=1, profile=False))
            r5 = box["r"]
            if "error" in r5:
                extra["config5"] = {"error": r5["error"], "transport": tr5}
            else:
                extra["config5"] = {
                    "workload": f"laplace3d-7pt {g5}^3 (n={g5 ** 3}, nnz={r5['nnz_global']}), nev={nev}, which=SR, mindim={mindim}, maxdim={maxdim}, "
                                f"rows/{world}; step = one Krylov-Schur restart cycle",
                    "metric": "arnoldi_iters_per_sec", "value": r5["state"]["steps"] / r5["elapsed"], "unit": "iters/s", "n_gpus": world,
                    "steps": r5["steps"], "warmup": r5["warmup"], "ms_per_step": 1e3 * r5["elapsed"] / max(r5["steps"], 1),
                    "transport": tr5, "rows_per_gpu": g5 ** 3 // world,
                    **(lambda tf: {"moved_GBps_per_gpu": tf["cycle_GBps"], "moved_frac": tf["cycle_frac"], "expand_frac": tf["expand_frac"]})(
                        traffic_fractions(r5["state"]["moved"], r5["state"].get("spec_bytes", 0.0), r5["state"]["t_expand"], r5["state"]["t_restart"], world, HBM_PEAK_GBS)),
                    "validation": r5["validation"],
                }
    if world == 1 and not args.no_shift_invert:
        try:
            # BASELINE config 4's size (n = 5e5) is the record; the n = 5e4 problem of round 3 rides beside it
            extra["shift_invert"] = shift_invert_record(pkg, 500, 1000, reps=10)
            try:
                extra["shift_invert"]["small"] = shift_invert_record(pkg, 200, 250, reps=20)
            except Exception as e:  # noqa: BLE001
                extra["shift_invert"]["small"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        except Exception as e:  # noqa: BLE001 - a side record must never cost the headline line
            extra["shift_invert"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    leave_last_words("cpu baseline / teardown")
    out = build_line()
    # Tear everything down first and flush the C stdio buffers (RCCL prints a version banner through printf,
    # which sits in the C buffer until exit when stdout is a pipe): the JSON line must be the LAST line on stdout.
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 - a broken transport must not eat the line
            pass
    if world > 1 and rank == 0:
        try:
            pkg._lib.load().ks_last_words(None, 0)  # the real line follows
        except Exception:  # noqa: BLE001
            pass
    emit(out)
    if out["value"] is None:
        sys.exit(3)
    if not out.get("validation", {}).get("ok", True) or not (out.get("config5") or {}).get("validation", {}).get("ok", True):
        sys.exit(4)  # the benched state violates the Arnoldi relation / orthogonality: the number is not a valid measurement


def spec_chain_products(nst, cap=10):
    """products of a speculative Newton chain for an expansion of nst steps (Float64; csrc/ks_backend.hpp: min(10, maxdim - k - 1))"""
    return max(0, min(cap, nst - 1))


def traffic_fractions(moved, spec_bytes, t_expand, t_restart, world, peak):
    """The traffic-true byte rates of the timed cycles, per GPU.  `cycle`: every byte the launched kernels moved over the WHOLE
    cycle time (expansion + restart interval) -- the figure to quote.  `expand`: the bytes executed inside the expansion intervals
    only, over those intervals: speculative products adopted by an expansion ran during the PREVIOUS restart interval, so they
    are taken out of the numerator (booking them to t_expand alone overstated round 5's figure: 0.652 against a true 0.545).
    (Cycles that are not synchronised one by one -- bench.py's default on one GPU --: an adopted chain may still be running when its
    expansion interval starts, so `expand` UNDER-states what ran inside the expansion intervals; `cycle` is exact either way.)"""
    t_cycle = max(t_expand + t_restart, 1e-12)
    cyc = moved / t_cycle / 1e9 / world
    exp = max(moved - spec_bytes, 0.0) / max(t_expand, 1e-12) / 1e9 / world
    return {"cycle_GBps": cyc, "cycle_frac": cyc / peak, "expand_GBps": exp, "expand_frac": exp / peak}


def make_line(args, pkg, passes, order, world, rank, force_dist, wl, with_cpu_baseline):
    """The ONE JSON line, from whatever passes completed."""
    m, n, nev, mindim, maxdim, which = wl["m"], wl["n"], wl["nev"], wl["mindim"], wl["maxdim"], wl["which"]
    valid = [t for t in order if t in passes and "error" not in passes[t]]
    # passes that completed must agree with each other (same restart trail, same Ritz values); the collective
    # transports (rccl, host) are the reference, a deviating peer-to-peer pass is dropped
    ref = next((t for t in ("rccl", "host") if t in valid), None)
    if ref is not None and "p2p" in valid:
        a, b = passes[ref]["state"], passes["p2p"]["state"]
        same = a["trail"] == b["trail"] and a["ritz"].shape == b["ritz"].shape and \
            float(np.abs(a["ritz"] - b["ritz"]).max()) <= 1e-8 * max(1.0, float(np.abs(a["ritz"]).max()))
        if not same:
            passes["p2p"] = {"error": f"results differ from the {ref} pass", **{k: v for k, v in passes["p2p"].items() if k == "elapsed"}}
            valid.remove("p2p")
    out = {
        "metric": "arnoldi_iters_per_sec", "value": None, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"laplace3d-7pt {m}^3 (n={n}), nev={nev}, which=SR, mindim={mindim}, maxdim={maxdim}, tol=sqrt(eps), "
                               f"explicit v1 (splitmix64 seed 20240917); step = one Krylov-Schur restart cycle",
                   "n": n, "parallelism": f"rows/{world}" if world > 1 else "single-gpu"},
    }
    if order != ["single"]:
        out["transports"] = {
            t: ({"error": p["error"]} if "error" in p else
                {"value": p["state"]["steps"] / p["elapsed"], "ms_per_step": 1e3 * p["elapsed"] / max(args.steps, 1)})
            for t, p in passes.items()
        }
    if not valid:
        return out
    chosen = min(valid, key=lambda t: passes[t]["elapsed"])
    elapsed, state, prof = passes[chosen]["elapsed"], passes[chosen]["state"], passes[chosen]["prof"]
    nnz_global, A_host, fmt = passes[chosen]["nnz_global"], passes[chosen]["A_host"], passes[chosen]["fmt"]
    bp = passes[chosen].get("basis_passes", 3)
    sst = passes[chosen].get("sstep", 0)
    out["value"] = state["steps"] / elapsed
    out["ms_per_step"] = 1e3 * elapsed / max(args.steps, 1)
    cyc = sorted(state.get("cycle_s", []))
    if cyc:   # spread of the timed cycles (the timed region is short: K cycles of a few ms)
        out["cycle_ms"] = {"min": 1e3 * cyc[0], "median": 1e3 * cyc[len(cyc) // 2], "max": 1e3 * cyc[-1], "n": len(cyc)}
    layout = fmt["layout"]
    out["config"].update({
        "workload": f"laplace3d-7pt {m}^3 (n={n}, nnz={nnz_global}), nev={nev}, which=SR, mindim={mindim}, maxdim={maxdim}, "
                    f"tol=sqrt(eps), explicit v1 (splitmix64 seed 20240917); step = one Krylov-Schur restart cycle",
        "nnz": nnz_global,
        "arnoldi_iterations_timed": state["steps"],
        "cycles_synchronised_one_by_one": bool(state.get("sync_cycles", True)),
        # per-step cycles: steps whose DGKS test asked for the second projection (src/expansion.jl:91).  Block cycles take no
        # such decision (the second stage is always part of a block): reported separately, not as DGKS passes
        "dgks_second_passes": state["reorth"] if not state.get("blk_cycles", 0) else None,
        "steps_in_blocks_with_second_stage": state["reorth"] if state.get("blk_cycles", 0) else None,
        "basis_passes_per_step": bp,  # 2: the DGKS second projection is carried in a triangular factor (implicit), 3: applied to the vector
        # s-step (block) expansion: steps per block; with it the basis is read twice per BLOCK (not per step)
        "sstep": {"s": sst, "block_cycles": state.get("blk_cycles", 0), "blocks": state.get("blk_blocks", 0),
                  "cycles_not_in_blocks": state.get("blk_irregular", 0), "smallest_pivot_ratio": state.get("blk_pivot_min"),
                  "largest_gram_deviation": state.get("blk_gram_dev"),
                  # block size in force at the end (the library halves it after an abandoned block) and abandoned blocks
                  "s_in_force": state.get("blk_s_now"), "abandoned_blocks": state.get("blk_abandoned"),
                  # cycles whose restart rotation ran fused with the first pass of the next block (its time and bytes are then
                  # part of expand_seconds / moved_bytes, not of restart_seconds)
                  "fused_rotations": state.get("fused_rotations", 0)} if sst else None,
        "spmv_layout": {"csr-dvi": "csr-dvi: %d-entry (column-row, value) dictionary, 1 B per non-zero (bit-identical products)",
                        "csr-vi": "csr-vi: %d-entry value dictionary, 4 B per non-zero (bit-identical products)",
                        "stencil": "stencil-mask: %d-slot (column-row, value) dictionary in the kernel arguments, 1 bit per slot and row (bit-identical products)",
                        "sell": "sell-64: sliced ELLPACK, 12 B per stored entry%.0s",
                        "sell-vi": "sell-64-vi: sliced ELLPACK with a %d-entry value dictionary, 4 B per stored entry",
                        "csr": "csr: 12 B per non-zero%.0s"}.get(layout, layout + "%.0s") % fmt["ndict"],
        "basis_placement": passes[chosen]["placement"],  # placement search of the workspace (off unless KS_PLACE_TRIALS > 1)
    })
    if chosen != "single":
        out["config"]["transport"] = chosen

    # ---- roofline ----
    t_cycle = max(state["t_expand"] + state["t_restart"], 1e-12)
    fused_gbs = state["bytes"] / t_cycle / 1e9 / world  # per GPU, over the whole cycle time like moved_*
    tf = traffic_fractions(state["moved"], state.get("spec_bytes", 0.0), state["t_expand"], state["t_restart"], world, HBM_PEAK_GBS)
    moved_gbs = tf["cycle_GBps"]   # whole-cycle rate: the quoted figure (moved_frac)
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
    if prof:
        # the second-pass update (class "axpy") skips itself on the device when the DGKS test did not ask for it
        # (src/expansion.jl:91); its bytes were booked at enqueue time -> scale by the fraction of steps that ran it
        if prof["axpy"]["count"]:
            prof["axpy"]["bytes"] *= min(1.0, state["reorth"] / max(1, state["steps"]))
        classes = {k: v for k, v in prof.items() if k != "fin" and v["count"] > 0}
        dom = max(classes, key=lambda k: classes[k]["ms"])
        d = classes[dom]
        ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        blk_dom = sst and state.get("blk_cycles", 0) and dom in BLOCK_KERNEL_NAMES and (dom != "rotate" or state.get("fused_rotations", 0))
        traffic = pmc_traffic(("blk_" + dom) if blk_dom else dom) if (m == 216 and world == 1 and not force_dist) else None
        if traffic is not None:
            traffic["measured_in_run"] = False  # read from the committed PMC passes (profiles/), not collected by this run
        roof.update({
            "kernel": ({k: v for k, v in BLOCK_KERNEL_NAMES.items() if k != "rotate" or state.get("fused_rotations", 0)}
                       if sst and state.get("blk_cycles", 0) else KERNEL_NAMES).get(dom, KERNEL_NAMES[dom]),
            "achieved": ach,
            "frac": ach / HBM_PEAK_GBS,
            "launches": d["count"],
            "avg_launch_ms": d["ms"] / d["count"],
            "algorithmic_bytes_per_launch": d["bytes"] / d["count"],
            "traffic": traffic,
            # the kernels of large blocks are not bound by memory alone: their FP64 vector rate beside the byte rate (flops of
            # an average launch of the dominant class x its launches / their time; peak: 78.6 TFLOP/s FP64 vector on MI355X)
            "fp64_vector": (lambda fl: {"achieved_TFLOPs": fl / 1e12, "peak_TFLOPs": 78.6, "frac": fl / 1e12 / 78.6})(
                state.get("blk_flops_" + dom, 0.0) / max(1, state.get("blk_blocks", 0)) * d["count"] / (d["ms"] * 1e-3) / world)
            if (sst and state.get("blk_cycles", 0) and dom in ("dots", "fused") and d["ms"] > 0) else None,
            "per_class": {k: {"ms_total": v["ms"], "launches": v["count"],
                              "GBps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 and v["bytes"] > 0 else None}
                          for k, v in prof.items()},
        })
        sp = prof.get("spmv")
        if sp and sp["count"] > 0 and sp["ms"] > 0:
            gbs = sp["bytes"] / (sp["ms"] * 1e-3) / 1e9
            # BASELINE.json's metric names "SpMV GB/s vs HBM peak": bytes = what THIS layout streams
            # (bytes_per_nnz * nnz + 4 (n+1) + 16 n) / HIP-event time; `csr_equivalent_GBps` prices the same
            # launch at the 12 B/nnz of plain CSR (SURVEY 8d's formula) -- a speed, not a traffic figure
            roof["spmv"] = {"layout": layout, "GBps": gbs, "frac": gbs / HBM_PEAK_GBS, "avg_launch_ms": sp["ms"] / sp["count"],
                            "bytes_per_launch": sp["bytes"] / sp["count"],
                            "csr_equivalent_GBps": (12.0 * nnz_global + 4.0 * (n + 1) + 16.0 * n) / world / (sp["ms"] / sp["count"] * 1e-3) / 1e9}
    else:
        roof.update({"kernel": "fused step (no per-kernel events)", "achieved": moved_gbs, "frac": moved_gbs / HBM_PEAK_GBS})
    roof["fused_step"] = {
        "what": ("S-STEP EXPANSION: per block of s steps, s operator products + two passes over the basis (k_bdots reads 8 n (k + s), "
                 "k_bupdate reads 8 n (k + s) and writes 8 n s; where the restart rotation ran fused with the first pass -- config.sstep.fused_rotations -- "
                 "that launch reads maxdim + 1 + s columns and writes the rotated ones, and belongs to the expansion); moved_* prices exactly those launches.  survey_compulsory_frac and "
                 "algorithmic_* price the same wall time with the PER-STEP byte counts of SURVEY 8d (two / four passes over V per step): "
                 "speeds relative to those op sequences, NOT bandwidths -- they exceed the HBM peak because the block form does not move "
                 "those bytes.  " if sst and state.get("blk_cycles", 0) else "") +
                "SpMV + DGKS per Arnoldi step over the expansion wall time, per GPU.  moved_*: bytes the launched kernels must "
                "move (" + ("two passes over V per step: the DGKS second projection is carried in a triangular factor, = SURVEY 8d's "
                            "compulsory B_step(j) for the layout in use; survey_compulsory_frac prices the same time with SURVEY's own formula, "
                            "12 B per non-zero of plain CSR, although the layout in use streams less" if bp == 2 else
                            "three passes over V when the second DGKS pass is taken") + ") over the WHOLE cycle time (expand_seconds + restart_seconds: "
                "speculative products of the next expansion run during the host step of a restart, so the two intervals share the device) "
                "-- the traffic-true figure, quote this one; expand_interval_*: the bytes executed inside the expansion intervals "
                "(speculative products taken out: spec_products of them ran in restart intervals) over expand_seconds; "
                "algorithmic_*: SURVEY 8d's formula for the UN-FUSED sequence with an explicit second pass (four passes over V) "
                "divided by the same time -- a speed relative to the reference's op sequence, NOT a bandwidth (it rewards fusion "
                "and the implicit second pass and can exceed the HBM peak)",
        "moved_bytes": state["moved"],
        "moved_GBps": moved_gbs,
        "moved_frac": moved_gbs / HBM_PEAK_GBS,
        "cycle_frac": tf["cycle_frac"],
        "moved_frac_of_measured_copy_ceiling": moved_gbs / 6290.0,
        "expand_interval_GBps": tf["expand_GBps"],
        "expand_interval_frac": tf["expand_frac"],
        "spec_products": state.get("spec_products", 0),
        "spec_bytes": state.get("spec_bytes", 0.0),
        "survey_compulsory_frac": (state.get("survey", 0.0) / t_cycle / 1e9 / world) / HBM_PEAK_GBS,
        "algorithmic_GBps": fused_gbs,
        "algorithmic_frac": fused_gbs / HBM_PEAK_GBS,
        "expand_seconds": state["t_expand"],
        "restart_seconds": state["t_restart"],
    }
    sc = passes[chosen].get("spmv_csr")
    if sc:
        tr_csr = pmc_traffic("spmv_csr") if (m == 216 and world == 1 and not force_dist) else None
        if tr_csr is not None:
            tr_csr["measured_in_run"] = False
        roof["spmv_plain_csr"] = dict(sc, what="the same matrix as plain CSR (12 B per non-zero) through k_spmv_csr: the SpMV "
                                               "north_star describes, measured in this run beside the layout the solver picked", traffic=tr_csr)
    out["roofline"] = roof
    if passes[chosen].get("validation") is not None:
        out["validation"] = passes[chosen]["validation"]

    # ---- CPU baseline: the reference's op sequence on the host cores (rank 0, N = 1 only) ----
    if with_cpu_baseline and world == 1 and rank == 0 and not args.no_cpu_baseline and A_host is not None:
        out["cpu_baseline"] = cpu_baseline(pkg, A_host, n, nev, which, mindim, maxdim)
    return out


def julia_runtests(julia, timeout=600):
    """First contact with the reference's own language: when the box has a `julia` that can load ArnoldiMethod, run
    arnoldimethod.jl_amd/julia/runtests.jl (the reference's test/expansion.jl and test/partial_schur.jl on a device basis through
    the reference's own partialschur!) and record what happened -- never part of the timed region, never fatal.  None: no Julia."""
    if not julia:
        return None
    import subprocess

    script = os.path.join(ROOT, "arnoldimethod.jl_amd", "julia", "runtests.jl")
    try:
        r = subprocess.run([julia, "--startup-file=no", script], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        return {"rc": r.returncode, "tail": (r.stdout + r.stderr)[-1500:]}
    except Exception as e:  # noqa: BLE001
        return {"rc": None, "tail": f"{type(e).__name__}: {e}"[:500]}


def cpu_baseline(pkg, A_host, n, nev, which, mindim, maxdim):
    """The reference's op sequence on the host cores (oracle/cpu_backend.cpp, kind "port": Julia is not in the image --
    `julia_on_box` records whether the GPU box has one).  Bounded samples of the same workload:
      value                     all cores for every verb (what a threaded BLAS + a threaded SpMV would give)
      serial_spmv               the same with ONE thread in the SpMV: Julia's SparseMatrixCSC mul! is serial
      scipy_eigs                scipy.sparse.linalg.eigs (ARPACK, implicitly restarted Arnoldi) on the same matrix, one
                                restart: operator applications per second, informational
      host_stream_triad_GBps    STREAM triad of the same OpenMP team: the ceiling of this bandwidth-bound baseline"""
    import shutil

    out = {"value": None, "unit": "iters/s", "cores": 0, "kind": "port", "sample": "", "julia_on_box": shutil.which("julia"),
           "logical_cpus": os.cpu_count(), "cpu_quota": CPU_BUDGET}
    out["julia_runtests"] = julia_runtests(out["julia_on_box"])
    try:
        from oracle import cpuref

        A = pkg.matrices.to_scipy(*A_host, n)
        tb = cpuref.timed_cycles_csr(A, nev=nev, which=which, mindim=mindim, maxdim=maxdim, cycles=3)
        out.update({
            "value": tb["steps"] / tb["seconds"],
            "cores": tb["threads"],
            "sample": f"same matrix and parameters; 3 restart cycles = {tb['steps']} Arnoldi iterations after the initial "
                      f"expansion (untimed), {tb['seconds']:.1f} s; un-fused reference op sequence with OpenMP over rows "
                      f"(spmv {tb['t_spmv']:.1f} s, orthogonalize {tb['t_orth']:.1f} s, rotation {tb['t_rot']:.1f} s)",
        })
        # bytes the un-fused sequence streams per step: 12 nnz + 4 (n+1) + 8 n (2 j + 9) + second pass 16 j n + 32 n
        jm = (mindim + maxdim) / 2 + 0.5
        step_b = 12.0 * A.nnz + 4.0 * (n + 1) + 8.0 * n * (2 * jm + 9) + 16.0 * jm * n + 32.0 * n
        out["effective_GBps"] = step_b * tb["steps"] / tb["seconds"] / 1e9
        try:
            out["host_stream_triad_GBps"] = cpuref.stream_triad_gbs()
        except Exception as e:  # noqa: BLE001
            out["host_stream_triad_GBps"] = f"failed: {e}"
        try:
            t1 = cpuref.timed_cycles_csr(A, nev=nev, which=which, mindim=mindim, maxdim=maxdim, cycles=1, spmv_threads=1)
            out["serial_spmv"] = {"value": t1["steps"] / t1["seconds"], "unit": "iters/s",
                                  "sample": f"1 restart cycle = {t1['steps']} iterations, {t1['seconds']:.1f} s (spmv {t1['t_spmv']:.1f} s on one thread)"}
        except Exception as e:  # noqa: BLE001
            out["serial_spmv"] = {"value": None, "sample": f"failed: {e}"}
        try:
            import scipy.sparse.linalg as spla

            cnt = [0]

            class _Enough(Exception):
                pass

            budget_s = 5.0  # bounded sample: ARPACK's first restart takes ~18 s on this matrix, a rate needs far less

            def mv(x):
                if time.perf_counter() - t0 > budget_s:
                    raise _Enough
                cnt[0] += 1
                return A @ x

            op = spla.LinearOperator(A.shape, matvec=mv, dtype=np.float64)
            t0 = time.perf_counter()
            try:
                spla.eigs(op, k=nev, which="SR", ncv=maxdim + 1, maxiter=1, tol=1e-8, v0=pkg.matrices.start_vector(n))
            except (spla.ArpackNoConvergence, _Enough):
                pass
            dt = time.perf_counter() - t0
            out["scipy_eigs"] = {"value": cnt[0] / dt, "unit": "operator applications/s",
                                 "sample": f"ARPACK dnaupd through scipy, ncv={maxdim + 1}, first restart cut off after {budget_s:.0f} s: "
                                           f"{cnt[0]} applications in {dt:.1f} s"}
        except Exception as e:  # noqa: BLE001
            out["scipy_eigs"] = {"value": None, "sample": f"failed: {e}"}
    except Exception as e:  # noqa: BLE001
        out["sample"] = f"failed: {e}"
    return out


if __name__ == "__main__":
    main()
