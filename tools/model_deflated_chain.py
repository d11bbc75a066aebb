#!/usr/bin/env python3
"""Model experiment (round 6): a Newton chain DEFLATED against locked Schur vectors inside the block.
Hypothesis for the blocks that :LM problems with dominant outliers abandon (open item 2): the chain
z_i = sigma (A - theta_i) z_{i-1} is orthogonalised against the basis only at the END of the block; for a non-normal A the product
A z has components along the locked Schur vectors U (coupling R12) even though z is orthogonal to them, and every later step
multiplies them by lambda_locked / |lambda_rest| (50 on the operator of test/partial_schur.jl:122-138): after 5-7 steps the
columns of the chain are parallel (pivot <= 0).  The per-step expansion never sees this -- it projects every vector at once.
Variant tested here (tests/sstep_model.py: expand_block2 patched):
    y = (A - theta_i) z_{i-1};   c_i = U^H y;   z_i = sigma (y - U c_i)         U = the locked columns (all, or the dominant ones)
    A z_{i-1} = z_i / sigma + theta_i z_{i-1} + U c_i                           -> c_i added to the locked rows in the H recovery
Prints products, blocks, abandoned blocks, block size in force at the end, cond(R_1), worst relation / orthogonality.
    python tools/model_deflated_chain.py            (~2 min)"""
import inspect
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import sstep_model as sm
from oracle import arnoldi as oa
from oracle.matrices import hashed_nonsymmetric

src = inspect.getsource(sm.expand_block2)
src = src.replace("def expand_block2(", "def expand_block2d(")
old = "            Z[:, i] = (apply(prev) - th[i] * prev) * sig[i]\n            prev = Z[:, i]\n"
new = ("            y_ = apply(prev) - th[i] * prev\n"
       "            if DEFL['U'] is not None:\n"
       "                c_ = DEFL['U'].conj().T @ y_\n"
       "                y_ = y_ - DEFL['U'] @ c_\n"
       "                DEFL['c'].append(c_)\n"
       "            Z[:, i] = y_ * sig[i]\n            prev = Z[:, i]\n")
assert old in src
src = src.replace(old, new)
old = "        a0 = zeta[:, 1] / sig[0] + th[0] * zeta[:, 0]\n"
new = old + "        if DEFL['U'] is not None:\n            a0[DEFL['rows']] += DEFL['c'][0]\n"
assert old in src
src = src.replace(old, new)
old = "            rhs = zeta[:, 2:] / sig[1:][None, :] + zeta[:, 1:s] * th[1:][None, :]\n"
new = old + "            if DEFL['U'] is not None:\n                for i_ in range(1, s):\n                    rhs[DEFL['rows'], i_ - 1] += DEFL['c'][i_]\n"
assert old in src
src = src.replace(old, new)
# one block per call of the patched function resets the coefficient list
src = src.replace("        Z = np.zeros((S.shape[0], s), dtype=dtype, order=\"F\")\n        prev = S[:, k - 1]\n",
                  "        Z = np.zeros((S.shape[0], s), dtype=dtype, order=\"F\")\n        prev = S[:, k - 1]\n        DEFL['c'] = []\n        DEFL['U'], DEFL['rows'] = pick_U(st, H, k)\n")
src = src.replace("        sig = np.full(s, scale if scale is not None else 1.0)\n",
                  "        sig = np.full(s, scale if scale is not None else 1.0)\n        if INTER['thr'] is not None:\n            sig = np.asarray([1.0 / _pow2(abs(t)) if abs(t) > INTER['thr'] else INTER['rest'] for t in th])\n")
exec(src, sm.__dict__)
sm.INTER = dict(thr=None, rest=1.0)
sm.DEFL = dict(U=None, c=[], rows=None)


def interleaved_shifts(ritz, s, real, every=2):
    """dominant Ritz values (|theta| > 4 x median) repeated: every `every`-th step shifts at one of them (the component of the chain
    along a dominant eigenvector is known to tol only -- a locked vector, a converged Ritz value -- and every other step multiplies it
    by |lambda_dom| / |lambda_rest|); per-step scale: 1 / |theta| for those steps, 1 / max|rest| otherwise"""
    r = np.asarray(ritz, dtype=np.complex128)
    a = np.abs(r)
    med = np.median(a)
    dom = r[a > 4.0 * med]
    rest = r[a <= 4.0 * med]
    if len(dom) == 0 or len(rest) == 0:
        sm.INTER['thr'] = None
        return sm.newton_shifts(ritz, s, real)
    dom_u = np.unique(np.round(dom.real, 14)) if real else dom
    lj = sm.newton_shifts(rest, s, real)
    out, di, ri = [], 0, 0
    for i in range(s):
        if i % every == 0:
            out.append(dom_u[di % len(dom_u)]); di += 1
        else:
            out.append(lj[ri % len(lj)]); ri += 1
    sm.INTER['thr'] = 2.0 * np.abs(rest).max()
    sm.INTER['rest'] = 1.0 / sm._pow2(np.abs(rest).max())
    return np.asarray(out).real.copy() if real else np.asarray(out)
MODE = dict(defl="none")


def pick_U(st, H, k):
    """locked columns = the leading columns whose sub-diagonal entry is exactly zero (src/run.jl:330 sets it); `dominant`: only
    those whose diagonal entry is at least 3x the median |diagonal| of the rest"""
    if MODE["defl"] == "none":
        return None, None
    if MODE["defl"] == "all-noshift":
        MODE_ = "all"
    nl = 0
    while nl < k - 1 and H[nl + 1, nl] == 0:
        nl += 1
    if nl == 0:
        return None, None
    rows = np.arange(nl)
    if MODE["defl"] == "dominant":
        d = np.abs(np.diag(H[:k - 1, :k - 1]))
        rest = np.median(d[nl:]) if k - 1 > nl else 0.0
        rows = rows[d[:nl] > 3.0 * rest]
        if len(rows) == 0:
            return None, None
    V = st.true_basis(k)
    return V[:, rows].copy(), rows
sm.pick_U = pick_U


def patched_expand(A, st, frm, to, stats, ritz, s, real, **kw):
    s = min(s, stats.get("s_eff", s))
    if s <= 1 or ritz is None:
        st.materialize(frm); sm.expand_steps(A, st, frm, to, stats); return
    rho = np.abs(np.asarray(ritz)).max(); scale = 1.0 / sm._pow2(max(rho, 1e-300))
    try:
        if MODE["defl"] == "all-noshift":
            # deflation + no shift at what is deflated: shifts (and the scale) from the Ritz values that are not locked
            sm.INTER['thr'] = None
            nl = 0
            while nl < frm - 2 and st.H[nl + 1, nl] == 0:
                nl += 1
            lockd = np.diag(st.H[:nl, :nl])
            rr = np.asarray([z for z in np.asarray(ritz) if not any(abs(z - l) <= 1e-8 * max(1.0, abs(l)) for l in lockd)])
            if len(rr) == 0:
                rr = np.asarray(ritz)
            scale = 1.0 / sm._pow2(max(np.abs(rr).max(), 1e-300))
            sh = sm.newton_shifts(rr, s, real)
        elif MODE.get("inter"):
            sh = interleaved_shifts(ritz, s, real, MODE["inter"])
        else:
            sm.INTER['thr'] = None
            sh = sm.newton_shifts(ritz, s, real)
        sm.expand_block2d(A, st, frm, to, sh, s, stats, scale=scale)
    except sm.BlockBail as b:
        stats["bails"] = stats.get("bails", 0) + 1
        stats["s_eff"] = s // 2 if s >= 4 else (2 if s > 2 else 1)
        st.materialize(b.step); sm.expand_steps(A, st, b.step, to, stats)
sm.expand = patched_expand


def run(A, label, s, nev, which, mind, maxd, defl, tol=1e-10, inter=0):
    MODE["defl"] = defl
    MODE["inter"] = inter
    label = label + (f" inter{inter}" if inter else "")
    n = A.shape[0]; v1 = oa.uniform_hash(20240917, np.arange(n))
    r = sm.solve(A, v1, nev, which, tol, mind, maxd, 100, np.float64, s=s)
    conds = [d[2] for d in r["diag"]] or [0]; gd = [d[4] for d in r["diag"]] or [0]
    ref, rh = oa.partialschur(A, v1=v1, nev=nev, which=which, tol=tol, mindim=mind, maxdim=maxd, restarts=100)
    print(f"{label:24s} s={s:2d} deflation={defl:9s}: prods {r['prods']} (oracle {rh.mvproducts}) blocks {r['stats'].get('blocks', 0)} bails {r['stats'].get('bails', 0)} "
          f"s_eff {r['stats'].get('s_eff')} cond med {np.median(conds):.1f} max {max(conds):.1e} gdev {max(gd):.1e} worst rel {r['worst']['rel']:.1e} orth {r['worst']['orth']:.1e}", flush=True)


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    n = 400
    D = rng.standard_normal((n, n)) / np.sqrt(n); D[0, 0] = 50.0      # test/partial_schur.jl:122-138: disc + outlier
    H3 = hashed_nonsymmetric(3000, seed=11, planted=[(3.0, 0.0), (2.5, 1.0), (-2.8, 0.0), (2.2, 0.7), (-2.0, 1.5), (1.9, 0.0)])
    if len(sys.argv) > 1 and sys.argv[1] == "defl":
        for defl in ("none", "all", "dominant"):
            run(D, "dense disc+outlier", 10, 5, "LM", 10, 30, defl)
            run(H3, "hashed nonsym n=3000", 10, 10, "LM", 10, 30, defl)
    if len(sys.argv) > 1 and sys.argv[1] == "noshift":
        cases = [(D, "dense disc+outlier", 10, 5, "LM", 10, 30), (H3, "hashed nonsym n=3000", 10, 10, "LM", 10, 30)]
        for c in cases:
            for defl in ("none", "all-noshift"):
                run(*c, defl)
        sys.exit(0)
    for inter in (0, 2, 3):
        run(D, "dense disc+outlier", 10, 5, "LM", 10, 30, "none", inter=inter)
        run(H3, "hashed nonsym n=3000", 10, 10, "LM", 10, 30, "none", inter=inter)
