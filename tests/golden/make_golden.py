"""Generates the committed golden fixtures under tests/golden/ from the ORACLE (oracle/arnoldi.py),
after it has been pinned to the reference's known-answer tests (tests/test_oracle_*.py).

The reference ships no golden vectors and cannot be run here (no Julia), so
these are regression/cross-implementation vectors: inputs and expected outputs only -- the H/Q pair
entering one restart, the grouping decisions and the H/Q pair leaving it, final eigenvalues and
mat-vec counts.  Run from the repo root:   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)

from oracle import arnoldi as oa  # noqa: E402
from oracle.matrices import hashed_nonsymmetric, laplace1d, laplace3d  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def case(name, A, **kw):
    n = A.shape[0]
    cplx = A.dtype.kind == "c"
    v1 = oa.uniform_hash(oa.DEFAULT_SEED, np.arange(n))
    if cplx:
        v1 = v1 + 1j * oa.uniform_hash(oa.DEFAULT_SEED + 1, np.arange(n))
    trace = []
    dec, hist = oa.partialschur(A, v1=v1, trace=trace, **kw)
    # keep up to 4 restarts spread over the run (first, two in the middle, last)
    idx = sorted(set([0, len(trace) // 3, 2 * len(trace) // 3, len(trace) - 1]))
    out = dict(
        eigenvalues=dec.eigenvalues, mvproducts=hist.mvproducts, nconverged=hist.nconverged, restarts=hist.restarts,
        nev=kw["nev"], mindim=kw.get("mindim", min(max(10, kw["nev"]), n)), maxdim=kw.get("maxdim", min(max(20, 2 * kw["nev"]), n)),
        tol=kw["tol"], which=kw["which"], n=n, steps=np.array(idx),
    )
    for t, i in enumerate(idx):
        tr = trace[i]
        # H entering the restart = Schur form already applied in trace; store the pre-Schur H by
        # undoing nothing: the oracle records H/Q after local_schurfact -> use H_in = Q H_schur Q' on
        # the leading block is lossy, so record the raw expansion output instead (see oracle trace).
        out[f"s{t}_active"] = tr["active"]
        out[f"s{t}_H_in"] = tr["H_in"]
        out[f"s{t}_k"] = tr["k"]
        out[f"s{t}_nlock"] = tr["nlock"]
        out[f"s{t}_purge"] = tr["purge"]
        out[f"s{t}_groups"] = tr["groups"]
        out[f"s{t}_lams"] = tr["lams"]
        out[f"s{t}_rs"] = tr["rs"]
        out[f"s{t}_H_after"] = tr["H_after"]
        out[f"s{t}_Q_after"] = tr["Q_after"]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, hist, "restarts kept:", idx)


if __name__ == "__main__":
    case("tridiag100_sr", laplace1d(100), nev=10, which="SR", tol=1e-6)
    case("laplace3d_8x9x10_sr", laplace3d(8, 9, 10), nev=6, which="SR", tol=1e-10, maxdim=30)
    planted = [(5.0, 3.0), (4.0, -2.5), (-6.0, 1.0), (7.5, 0.0)]
    case("nonsym300_lm", (hashed_nonsymmetric(300, seed=7, planted=planted) * 1.0).tocsr(), nev=6, which="LM", tol=1e-10)
    import scipy.sparse as sp

    d = np.arange(1, 81) * (1 + 0.25j)
    Ac = (sp.diags(d) + 0.01 * (hashed_nonsymmetric(80, seed=11) + 1j * hashed_nonsymmetric(80, seed=12))).tocsr()
    case("complex80_lr", Ac, nev=5, which="LR", tol=1e-10)
