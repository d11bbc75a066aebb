"""`-m "not gpu"`: the N > 1 path on CPU -- world_size 2 (and 3) `gloo` process groups exercise the row
partition, the halo plan handed to ks_operator_csr_dist and the sharded DGKS expansion with
all-reduced coefficients (tests/dist_worker.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from __graft_entry__ import ROOT, import_package

pkg = import_package()
from arnoldimethod_jl_amd import dist as ksd  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_path_under_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DIST_WORKER_OK" in r.stdout


def test_partition_rows():
    offs = ksd.partition_rows(216 ** 3, 8, granule=216 * 216)
    assert offs[0] == 0 and offs[-1] == 216 ** 3 and (np.diff(offs) == 27 * 216 * 216).all()
    offs = ksd.partition_rows(10, 3)
    assert offs.tolist() == [0, 4, 7, 10]
    with pytest.raises(AssertionError):
        ksd.partition_rows(10, 2, granule=4)


def test_plan_without_process_group_single_rank():
    ip, ix, dv = pkg.matrices.laplace3d_csr(4, 4, 4, index_dtype=np.int64)
    plan = ksd.build_halo_plan(ix, np.array([0, 64]), 0)
    assert plan.nghost == 0 and len(plan.neigh) == 0 and (plan.colidx_local == ix).all()


def test_plan_two_ranks_in_process():
    """Both ranks' plans built in one process through the `exchange` hook: structure of the slab plan."""
    m = 5
    n = m ** 3
    offs = ksd.partition_rows(n, 2, granule=m * m)
    blocks = [pkg.matrices.laplace3d_csr(m, m, m, int(offs[r]), int(offs[r + 1]), index_dtype=np.int64) for r in range(2)]
    needs = {}
    for r in range(2):
        idx = blocks[r][1]
        r0, r1 = offs[r], offs[r + 1]
        g = np.unique(idx[(idx < r0) | (idx >= r1)])
        owner = np.searchsorted(offs, g, side="right") - 1
        needs[r] = [g[owner == q] for q in range(2)]
    plans = [ksd.build_halo_plan(blocks[r][1], offs, r, exchange=lambda need: [needs[0], needs[1]]) for r in range(2)]
    for r, pl in enumerate(plans):
        assert pl.neigh.tolist() == [1 - r] and pl.nghost == m * m and pl.recv_cnt.tolist() == [m * m]
        assert pl.send_ptr.tolist() == [0, m * m]
    # rank 0 sends its LAST plane, rank 1 its FIRST plane
    assert plans[0].send_idx.tolist() == list(range(int(offs[1]) - m * m, int(offs[1])))
    assert plans[1].send_idx.tolist() == list(range(m * m))
