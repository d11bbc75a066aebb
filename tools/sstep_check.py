"""Development check of the s-step (block) expansion on the GPU: lockstep of a per-step workspace and a block workspace.
   python tools/sstep_check.py [s ...]
Both workspaces run the same first expansion and the same restart (bit-identical states), then the second expansion goes
step by step in one and in blocks in the other: H columns, V (materialised), the device-side invariants."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import import_package  # noqa: E402

ks = import_package()
from oracle.matrices import laplace3d  # noqa: E402
from oracle import arnoldi as oa  # noqa: E402


def lockstep(A, dtype, s, nev, mindim, maxdim, which, cycles=3, tol=1e-10):
    n = A.shape[0]
    v1 = oa.uniform_hash(3, np.arange(n)).astype(dtype)
    if np.dtype(dtype).kind == "c":
        v1 = v1 + 1j * oa.uniform_hash(4, np.arange(n))
    op = ks.csr_operator(A)
    out = []
    wss = []
    for sstep in (0, s):
        ws = ks.ArnoldiWorkspace(n, maxdim, dtype)
        ws.set_sstep(sstep)
        ws.reinitialize(0, v1)
        ws.iterate_arnoldi(op, 1, mindim)
        wss.append(ws)
    k = [mindim, mindim]
    active = [0, 0]
    for cyc in range(cycles):
        Hs, Vs = [], []
        for w, ws in enumerate(wss):
            st = ws.iterate_arnoldi(op, k[w] + 1, maxdim)
            rel, orth = ws.arnoldi_relation(op, maxdim)
            H = np.array(ws.H)
            Hs.append(H)
            Vs.append(np.array(ws.V))
            info = ws.sstep_info
            print(f"  cycle {cyc} sstep={info['s']}: k={k[w]} steps={st['steps']} rel={rel:.2e} (|H|={np.linalg.norm(H):.2e}) orth={orth:.2e} "
                  f"blocks={info['blocks']} abandoned={info['abandoned']} piv1={info['pivot_stage1']:.2e} piv2={info['pivot_stage2']:.2e} gdev={info['gram_dev']:.2e}")
        dH = np.abs(Hs[0] - Hs[1]).max() / np.abs(Hs[0]).max()
        dV = np.abs(Vs[0] - Vs[1]).max()
        print(f"  cycle {cyc}: max |H_step - H_block| / |H| = {dH:.2e}   max |V_step - V_block| = {dV:.2e}")
        out.append((dH, dV))
        for w, ws in enumerate(wss):
            r = ws.restart(active[w], nev, which, tol, mindim, maxdim)
            k[w], active[w] = r["k"], min(r["nlock"], nev - 1)
        assert k[0] == k[1], k
    return out


if __name__ == "__main__":
    ss = [int(a) for a in sys.argv[1:]] or [2, 5]
    for s in ss:
        print(f"== laplace 20x21x22 float64 s={s}")
        lockstep(laplace3d(20, 21, 22), np.float64, s, 20, 20, 40, "SR")
        print(f"== nonsymmetric n=4000 float64 s={s}")
        A = (sp.random(4000, 4000, density=5.0 / 4000, random_state=np.random.default_rng(3), format="csr") + sp.diags(np.linspace(1, 3, 4000))).tocsr()
        lockstep(A, np.float64, s, 8, 10, 20, "LM")
        if s <= 5:
            print(f"== complex n=990 s={s}")
            Ac = (laplace3d(9, 10, 11) + 1j * sp.diags(0.3 * np.cos(np.arange(990)))).tocsr().astype(np.complex128)
            lockstep(Ac, np.complex128, s, 6, 10, 20, "LM")
    # whole solves
    A = laplace3d(14, 15, 16)
    n = A.shape[0]
    v1 = oa.uniform_hash(3, np.arange(n))
    kw = dict(nev=20, which="SR", tol=1e-8, mindim=20, maxdim=40, restarts=200)
    ref, rh = oa.partialschur(A, v1=v1, **kw)
    for s in [0] + ss:
        os.environ["KS_SSTEP"] = str(s)
        t = time.time()
        dec, hist = ks.partialschur(A, v1=v1, **kw)
        Q, R = dec.Q, np.array(dec.R)
        print(f"solve sstep={s}: {hist} oracle mvproducts={rh.mvproducts} res={np.linalg.norm(A @ Q - Q @ R):.2e} orth={np.linalg.norm(Q.T @ Q - np.eye(Q.shape[1])):.2e} "
              f"eigdiff={np.abs(np.sort(dec.eigenvalues.real) - np.sort(ref.eigenvalues.real)).max():.2e} [{time.time() - t:.2f} s]")
