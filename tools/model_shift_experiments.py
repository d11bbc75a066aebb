#!/usr/bin/env python3
"""Model experiments on the Newton shifts of the block expansion for NON-symmetric real operators (round 5, review item 6).
Runs tests/sstep_model.py (numpy statement of the block algorithm, pinned to the oracle) with patched shift rules on
 (a) a dense random matrix with a dominant outlier, the operator of test/partial_schur.jl:122-138 (spectrum {50} + disc of radius 1),
 (b) the hashed nonsymmetric matrix of config 3 in miniature,
and prints products, blocks, abandoned blocks, block size in force at the end, cond(R_1) and the worst Arnoldi relation:
   base      real parts of the Ritz values, one scale 1 / max|theta|                    (what the library does)
   pairs     complex-conjugate shift pairs in real arithmetic, z_i = s (A - a) z_{i-1} + s^2 b^2 z_{i-2} for theta = a +- i b,
             H recovered from the three-term recurrence
   perstep   per-step scale: 1 / |theta| for an outlier shift, 1 / (largest of the rest) otherwise
Outcome (python tools/model_shift_experiments.py, ~2 min): on (a) all three abandon the same blocks at the first restarts
(cond 2.6e3, pivot <= 0 at the 5th-7th column) and end with blocks of 1-2; on (b) pairs change cond(R_1) from 7.6e4 to 1.1e5.
Conjugate pairs are NOT what these problems lack -- not built on the device."""
import sys, inspect
import os
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import sstep_model as sm
from oracle import arnoldi as oa
from oracle.matrices import hashed_nonsymmetric

def pair_shifts(ritz, s):
    """Leja order over the points with Im >= 0 (pairs weighted twice), expanded: real -> one step, complex -> two consecutive steps"""
    r = np.asarray(ritz, dtype=np.complex128)
    r = r[np.isfinite(r)]
    up = []
    for z in r:
        if z.imag < -1e-14*max(1,abs(z)): continue
        z = complex(z.real, z.imag if abs(z.imag) > 1e-14*max(1,abs(z)) else 0.0)
        if not any(abs(z-q) <= 1e-14*max(1,abs(z)) for q in up): up.append(z)
    if not up: return [('r',0.0,0.0)]*s
    # Leja with conjugates counted
    out=[]; pts=list(up)
    cur = int(np.argmax(np.abs(pts))); logp=np.zeros(len(pts)); used=[False]*len(pts)
    while len(out) < len(pts):
        used[cur]=True; out.append(pts[cur])
        best=None
        for i,z in enumerate(pts):
            if used[i]: continue
            logp[i] += np.log(max(abs(z-pts[cur]),1e-300)) + (np.log(max(abs(z-np.conj(pts[cur])),1e-300)) if pts[cur].imag else 0)
            if best is None or logp[i] > logp[best]: best=i
        if best is None: break
        cur=best
    seq=[]
    i=0
    while len(seq) < s:
        z = out[i % len(out)]; i+=1
        if z.imag == 0: seq.append(('r', z.real, 0.0))
        else:
            if len(seq)+2 <= s: seq += [('p1', z.real, z.imag), ('p2', z.real, z.imag)]
            else: seq.append(('r', z.real, 0.0))
    return seq

src = inspect.getsource(sm.expand_block2)
src = src.replace("def expand_block2(", "def expand_block2p(")
src = src.replace("        th = np.asarray([shifts[(blk * s_max + i) % len(shifts)] for i in range(s)], dtype=dtype)\n",
 "        kinds = [shifts[i] for i in range(s)]\n        if kinds[s-1][0]=='p1': kinds[s-1]=('r',kinds[s-1][1],0.0)\n        th = np.asarray([q[1] for q in kinds], dtype=dtype)\n        gam = np.asarray([-(scale if scale is not None else 1.0)*q[2]**2 if q[0]=='p2' else 0.0 for q in kinds], dtype=dtype)\n")
src = src.replace("            Z[:, i] = (apply(prev) - th[i] * prev) * sig[i]\n            prev = Z[:, i]\n",
 "            Z[:, i] = (apply(prev) - th[i] * prev) * sig[i]\n            if gam[i] != 0: Z[:, i] -= sig[i] * gam[i] * (Z[:, i-2] if i >= 2 else S[:, k-1])\n            prev = Z[:, i]\n")
# rhs: add gamma term: A z_{i-1} = z_i/sig + th z_{i-1} + gam_i z_{i-2};  columns i = 2..s (index into zeta: z_j <-> zeta[:, j])
src = src.replace("            rhs = zeta[:, 2:] / sig[1:][None, :] + zeta[:, 1:s] * th[1:][None, :]\n",
 "            rhs = zeta[:, 2:] / sig[1:][None, :] + zeta[:, 1:s] * th[1:][None, :] + zeta[:, 0:s-1] * gam[1:][None, :]\n")
exec(src, sm.__dict__)
src2 = inspect.getsource(sm.expand_block2)
src2 = src2.replace("        sig = np.full(s, scale if scale is not None else 1.0)\n",
 "        sig = np.full(s, scale if scale is not None else 1.0)\n        if PERSTEP is not None:\n            sig = np.asarray([ (1.0/_pow2(abs(t)) if abs(t) > PERSTEP[1] else PERSTEP[0]) for t in th ])\n")
exec(src2, sm.__dict__)
sm.PERSTEP=None

MODE = dict(pairs=False, trim=False)
def patched_expand(A, st, frm, to, stats, ritz, s, real, **kw):
    s = min(s, stats.get("s_eff", s))
    if s <= 1 or ritz is None:
        st.materialize(frm); sm.expand_steps(A, st, frm, to, stats); return
    rho = np.abs(np.asarray(ritz)).max(); scale = 1.0 / sm._pow2(max(rho, 1e-300))
    if MODE['trim']:
        a = np.sort(np.abs(np.asarray(ritz)))
        rest = a[a < 0.5 * a.max()]
        if len(rest) >= 4:
            sm.PERSTEP = (1.0 / sm._pow2(rest.max()), 2.0 * rest.max())
        else:
            sm.PERSTEP = None
    else:
        sm.PERSTEP = None
    try:
        if MODE['pairs']:
            sm.expand_block2p(A, st, frm, to, pair_shifts(ritz, s), s, stats, scale=scale)
        else:
            sm.expand_block2(A, st, frm, to, sm.newton_shifts(ritz, s, real), s, stats, scale=scale)
    except sm.BlockBail as b:
        print("   bail:", b.args, "s=", s, "frm", frm, "ritz", np.round(np.sort_complex(np.asarray(ritz))[-6:],3))
        stats["bails"] = stats.get("bails", 0) + 1
        stats["s_eff"] = s // 2 if s >= 4 else (2 if s > 2 else 1)
        st.materialize(b.step); sm.expand_steps(A, st, b.step, to, stats)
sm.expand = patched_expand

def run(A, label, s, nev, which, mind, maxd, pairs, trim=False):
    MODE['pairs']=pairs; MODE['trim']=trim
    n=A.shape[0]; v1 = oa.uniform_hash(20240917, np.arange(n))
    r = sm.solve(A, v1, nev, which, 1e-10, mind, maxd, 100, np.float64, s=s)
    conds=[d[2] for d in r["diag"]] or [0]; gd=[d[4] for d in r["diag"]] or [0]
    ref, rh = oa.partialschur(A, v1=v1, nev=nev, which=which, tol=1e-10, mindim=mind, maxdim=maxd, restarts=100)
    print(f"{label:26s} s={s:2d} pairs={pairs}: prods {r['prods']} (oracle {rh.mvproducts}) blocks {r['stats'].get('blocks',0)} bails {r['stats'].get('bails',0)} s_eff {r['stats'].get('s_eff')} cond med {np.median(conds):.1f} max {max(conds):.1e} gdev {max(gd):.1e} worst {r['worst']}")

rng=np.random.default_rng(5)
n=400
D = rng.standard_normal((n,n))/np.sqrt(n); D[0,0]=50.0      # like test/partial_schur.jl:122-138: disc + outlier
import scipy.sparse as sp
H3 = hashed_nonsymmetric(3000, seed=11, planted=[(3.0,0.0),(2.5,1.0),(-2.8,0.0),(2.2,0.7),(-2.0,1.5),(1.9,0.0)])
for pairs in (False, True):
    run(D, "dense disc+outlier", 10, 5, "LM", 10, 30, pairs)
    run(H3, "hashed nonsym n=3000", 10, 10, "LM", 10, 30, pairs)
run(D, "dense disc+outlier perstep", 10, 5, "LM", 10, 30, False, True)
run(H3, "hashed nonsym perstep", 10, 10, "LM", 10, 30, False, True)
