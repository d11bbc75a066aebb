"""Probe: the ill-posed cases of tests/test_gpu_factored_basis_stress.py by block size / kernel family."""
import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import import_package
pkg = import_package()
from test_gpu_factored_basis_stress import _ill_posed_case, _residual
for seed in (int(x) for x in (sys.argv[1:] or ["1", "3", "5"])):
    A, v1, kw = _ill_posed_case(seed)
    nb = sp.linalg.norm(A)
    dec, h = pkg.partialschur(A, v1=v1, **kw)
    res, orth = _residual(A, dec, h.nconverged)
    print(f"seed {seed} KS_SSTEP={os.environ.get('KS_SSTEP')} KS_BLK_RING={os.environ.get('KS_BLK_RING')}: res {res / nb:.2e} orth {orth:.1e} locked {h.nconverged} products {h.mvproducts} {h}", flush=True)
