"""`-m gpu`: bench.py must print its ONE JSON line whatever happens to a transport ("make the N > 1 bench
un-killable").  Two ranks share device 0 (KS_SAME_DEVICE=1: peer-to-peer + host-staged transports, RCCL refuses two
ranks per device); one pass is made to fail, or to hang, on purpose."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench(extra_env, nproc=2, grid=64, timeout=420, extra_args=()):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, KS_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--grid", str(grid), *extra_args]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    return r, json.loads(lines[0])


def test_both_transports_measured_and_agree():
    r, d = _bench({})
    assert r.returncode == 0
    assert set(d["transports"]) == {"p2p", "host"} and all("value" in v for v in d["transports"].values()), d["transports"]
    assert d["value"] == max(v["value"] for v in d["transports"].values()) and d["n_gpus"] == 2
    assert d["roofline"]["fused_step"]["moved_frac"] > 0 and d["scaling"] == "strong"
    # the collective-structured transport is measured FIRST (on an 8-GPU node: RCCL, alone, before any peer-to-peer set-up)
    assert list(d["transports"])[0] == "host"
    # the line validates the benched state (test/expansion.jl:29-30 on the device)
    v = d["validation"]
    assert v["ok"] and v["arnoldi_rel"] <= v["limit_rel"] and v["orth"] <= v["limit_orth"], v


def test_config5_record_rides_in_the_same_line():
    """BASELINE config 5 (464^3 over the ranks) as a second record of the N > 1 line -- here at 48^3 over 2 ranks on one GPU."""
    r, d = _bench({}, extra_args=("--config5", "--config5-grid", "48"))
    assert r.returncode == 0, r.stderr[-2000:]
    c5 = d["config5"]
    assert c5["n_gpus"] == 2 and c5["value"] > 0 and c5["steps"] == 3 and "48^3" in c5["workload"]
    assert c5["transport"] == d["config"]["transport"] and c5["validation"]["ok"], c5
    assert d["value"] > 0 and "64^3" in d["config"]["workload"]


def test_single_gpu_line_validates_itself_and_reports_the_plain_csr_spmv():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--grid", "96", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=420)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    v = d["validation"]
    assert v["ok"] and v["arnoldi_rel"] < 1e-11 and v["orth"] < 1.5e-10, v
    pc = d["roofline"]["spmv_plain_csr"]
    assert pc["layout"] == "csr" and pc["bytes_per_nnz"] == 12.0 and pc["measured_in_run"] and pc["launches"] == 20 and pc["GBps"] > 0, pc
    assert d["roofline"]["spmv"]["layout"] == "stencil"
    # the side record of the sparse shift-invert operator (BASELINE config 4 in its general form), measured in the run
    si = d["shift_invert"]
    assert si["measured_in_run"] and si["repeatable"] and si["max_rel_diff_vs_host_solve"] < 1e-11 and 0.0 < si["ms_per_product"] < si["host_solve_ms"], si


@pytest.mark.parametrize("victim,survivor", [("host", "p2p"), ("p2p", "host")])
def test_line_survives_a_failing_transport(victim, survivor):
    r, d = _bench({"KS_BENCH_INJECT_FAIL": victim})
    assert r.returncode == 0
    assert "error" in d["transports"][victim] and "value" in d["transports"][survivor]
    assert d["config"]["transport"] == survivor and d["value"] == d["transports"][survivor]["value"]


def test_line_survives_a_hung_transport():
    r, d = _bench({"KS_BENCH_TRANSPORTS": "host,p2p", "KS_BENCH_INJECT_FAIL": "p2p:hang", "KS_BENCH_PASS_DEADLINE_S": "20"})
    assert "did not finish" in d["transports"]["p2p"]["error"]
    assert d["value"] == d["transports"]["host"]["value"] and d["value"] > 0


def test_line_appears_even_when_nothing_survives():
    r, d = _bench({"KS_BENCH_TRANSPORTS": "p2p", "KS_BENCH_INJECT_FAIL": "p2p"})
    assert r.returncode != 0 and d["value"] is None and "error" in d["transports"]["p2p"]


def test_line_survives_a_rank_that_dies_the_hard_way():
    """First contact of a transport with a new fabric can end in a memory fault, and the GPU runtime then ABORTS the process:
    no exception, no watchdog.  Rank 1 aborts inside the second pass; the launcher tears rank 0 down with SIGTERM; the line
    built from the pass that had completed is what the library writes on rank 0's way out (ks_last_words)."""
    r, d = _bench({"KS_BENCH_TRANSPORTS": "host,p2p", "KS_BENCH_INJECT_FAIL": "p2p:crash"})
    assert d["died_during"] == "p2p" and "value" in d["transports"]["host"], d
    assert d["value"] == d["transports"]["host"]["value"] and d["value"] > 0 and d["config"]["transport"] == "host"


def test_eight_ranks_end_to_end_on_one_device():
    """The driver's scaling run is `bench.py --gpus 8` under torch.distributed.run; no 8-GPU node has been available to any
    round so far, so its first contact with one must not be the first run of that code path: the same command
    line with all eight ranks on device 0 (peer-to-peer and host-staged transports; RCCL refuses eight ranks on one device), the
    default block expansion, config 5's record riding along.  Asserted: ONE well-formed line, n_gpus 8, every rank's share in
    the validation, the roofline block and both transports present."""
    r, d = _bench({}, nproc=8, grid=64, timeout=900, extra_args=("--config5", "--config5-grid", "48"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["n_gpus"] == 8 and d["value"] > 0 and d["steps"] == 3 and d["scaling"] == "strong" and d["config"]["parallelism"] == "rows/8"
    # the collective-structured (host-staged) pass must succeed.  The peer-to-peer pass spins inside kernels for its peers:
    # with EIGHT processes time-sliced on one device a peer's kernel is sometimes not scheduled within the 30-s clock (measured:
    # 1 run in 4 ends in CommTimeout here, none with 2-4 ranks; one rank per GPU has no such co-scheduling) -- the line must
    # then carry that error and come from the surviving pass, which is the very property this file tests
    assert set(d["transports"]) == {"p2p", "host"} and "value" in d["transports"]["host"], d["transports"]
    assert "value" in d["transports"]["p2p"] or "CommTimeout" in d["transports"]["p2p"]["error"], d["transports"]
    v = d["validation"]
    assert v["ok"] and v["arnoldi_rel"] <= v["limit_rel"] and v["orth"] <= v["limit_orth"], v
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["fused_step"]["moved_frac"] > 0
    assert d["config"]["sstep"]["s"] == 20 and d["config"]["sstep"]["block_cycles"] > 0, d["config"]["sstep"]
    # config 5's record runs on the winning transport: when that is the peer-to-peer one, the same co-scheduling limit applies
    c5 = d["config5"]
    if "error" in c5:
        assert c5.get("transport") == "p2p" and "CommTimeout" in c5["error"], c5
    else:
        assert c5["n_gpus"] == 8 and c5["value"] > 0 and c5["validation"]["ok"], c5
