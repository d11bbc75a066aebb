#!/usr/bin/env python3
"""Model experiments (round 6) on the DRIFT of the Arnoldi relation under large blocks and on block TRUNCATION, run on
tests/sstep_model.py (the numpy statement of the block algorithm, pinned to the oracle) with patched policies.  CPU only, ~3 min.
    python tools/model_drift_experiments.py segments    relation residual per column segment (kept | block 1 | block 2 ...) per cycle,
                                                        seed 17 of tests/test_gpu_random_stress.py, blocks of 10 and of 8
    python tools/model_drift_experiments.py amp         ||PC R^-1||_F per block on the 40 first seeds + two healthy operators: is the
                                                        amplification of the H recovery a usable discriminator?  (no)
    python tools/model_drift_experiments.py watch       a watch that extrapolates the measured growth and HALVES the block size
    python tools/model_drift_experiments.py truncate    a block truncated at the first bad pivot / cancelling column instead of abandoned
    python tools/model_drift_experiments.py repair      the drift split into its part inside / outside span(V); H recomputed (= V^H A V) before the restart
Outcome (profiles/r06_model_drift_experiments.txt):
  * the drift is built INSIDE a cycle, block after block: the later blocks of a cycle (k = 34, 44 of 58) multiply the residual of the
    columns they lean on by 20-75, the restart hands the maximum to the next cycle; blocks of 8 on the same operator stay at 1e-14;
  * ||PC R^-1|| does not separate the cases: 4.5 (median) on the Laplacian, 490 on the hashed matrix of config 3 (relation 1e-11,
    flat), 5-10 on seed 17 BEFORE the onset, 18-79 at it -- a cap derived from it is 'blocks of 8 always' on seed 17 (which works
    there: 8e-14 over 40 cycles) and would cut config 3's blocks for nothing;
  * growth extrapolation reacts one to two cycles too late: x93 per cycle from 2e-12 means 1e-9 before blocks of 5 are in force,
    and the error stays (nothing in a Krylov-Schur cycle shrinks the residual of the kept columns);
  * truncation: on the dominant-outlier operators the FIRST or second chain vector already cancels (written block 0.96 away from
    orthonormal at k = 12): nothing to truncate to; on the planted config-3 miniature one block of 10 becomes 8-9 + a wasted product.
  * repair (round 6c): E = A V - V H has a component OUTSIDE span(V) as large as the one inside (3e-9 | 3e-9 at the onset): the drift is
    not an error of the recovered H alone; recomputing H = V^H A V before the restart stops the growth (5e-9, flat for 30 cycles) but
    does not bring the relation back.
None of these went to the device.  What stands: detection (drift watch) + step-by-step fallback; open item in DESIGN section 9."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SRC = open(os.path.join(ROOT, "tests", "sstep_model.py")).read()
REL = 'worst["rel"] = max(worst["rel"], float(np.linalg.norm(A @ V[:, :maxdim] - V @ H) / np.linalg.norm(H)))'
SLOOP = "        s = min(s_max, to - j + 1)\n        k = j\n"
ZETA = "        zeta = np.zeros((m, s + 1), dtype=dtype)\n        zeta[:k, 0] = u\n        zeta[:k, 1:] = PC"


def case(seed):
    from test_gpu_random_stress import _case
    return _case(seed)


def run(src, A, v1, kw, s, restarts=None, **glob):
    ns = dict(glob)
    exec(compile(src, "sstep_model_patched", "exec"), ns)
    return ns["solve"](A, v1, kw["nev"], kw["which"], kw["tol"], kw["mindim"], kw["maxdim"], restarts or kw["restarts"], A.dtype, s=s)


def segments():
    A, v1, kw, _ = case(17)
    src = SRC.replace(REL, 'worst["rel"] = 0.0; _E = A @ V[:, :maxdim] - V @ H; stats.setdefault("colres", []).append(np.linalg.norm(_E, axis=0) / np.linalg.norm(H)); stats.setdefault("ks", []).append(k)')
    for s in (10, 8):
        st = run(src, A.toarray(), v1, kw, s, restarts=12)["stats"]
        print(f"seed 17, blocks of {s}: max relative residual of the columns of each segment (kept | blocks in order), per cycle")
        for c, (cr, k) in enumerate(zip(st["colres"], st["ks"])):
            segs, j = [(0, k)], k
            while j < kw["maxdim"]:
                segs.append((j, min(j + s, kw["maxdim"])))
                j += s
            print("  cycle %2d  k=%d   " % (c, k) + "  ".join("%.0e" % cr[a:b].max() for a, b in segs))


def repair():
    """Where does the drift live?  E = A V_m - V_{m+1} H split into its part INSIDE span(V_{m+1}) (V^H E: an error of H alone, which a
    recomputed H = V^H A V removes) and OUTSIDE it ((I - V V^H) E: the space is not a Krylov space any more).  Then the same run
    with H REPAIRED (H += V^H E) before the restart whenever the relation exceeds 1e-12 ||H||."""
    A, v1, kw, _ = case(17)
    probe = ('_E = A @ V[:, :maxdim] - V @ H; _in = V.conj().T @ _E; _out = _E - V @ _in; _hn = np.linalg.norm(H); '
             'stats.setdefault("rels", []).append((np.linalg.norm(_E) / _hn, np.linalg.norm(_in) / _hn, np.linalg.norm(_out) / _hn)); worst["rel"] = 0.0')
    fix = probe + '\n            if REPAIR and np.linalg.norm(_E) > REPAIR * _hn:\n                H[:, :] += _in; stats["repairs"] = stats.get("repairs", 0) + 1'
    for s_, thr in ((10, 0.0), (10, 1e-12), (10, 1e-13), (20, 1e-12)):
        src = SRC.replace(REL, fix)
        r = run(src, A.toarray(), v1, kw, s_, restarts=40, REPAIR=thr)
        st = r["stats"]
        print(f"seed 17, blocks of {s_}, repair above {thr:g}: repairs {st.get('repairs', 0)}, cycles {len(st['rels'])}, converged {len(r['eig'])} of {kw['nev']}, products {r['prods']}")
        print("   relation | inside span | outside span, per cycle: " + "  ".join("%.0e|%.0e|%.0e" % t for t in st["rels"][:24]))


def amp():
    src = SRC.replace(REL, REL + '; stats.setdefault("rels", []).append(float(np.linalg.norm(A @ V[:, :maxdim] - V @ H) / np.linalg.norm(H)))')
    src = src.replace(ZETA, "        stats.setdefault('amp', []).append(float(np.linalg.norm(PC @ np.linalg.inv(R), 'fro')))\n" + ZETA)
    from oracle.matrices import hashed_nonsymmetric, laplace3d
    rng = np.random.default_rng(0)
    L = laplace3d(20, 21, 22).toarray()
    B = hashed_nonsymmetric(3000, seed=7).toarray()
    for name, A, kw in (("laplace 20x21x22 nev 20 SR 20/40", L, dict(nev=20, which="SR", tol=1e-10, mindim=20, maxdim=40, restarts=40)),
                        ("hashed 3000 nev 10 LM 10/20", B, dict(nev=10, which="LM", tol=1e-10, mindim=10, maxdim=20, restarts=40))):
        st = run(src, A, rng.standard_normal(A.shape[0]), kw, 20)["stats"]
        print(f"{name}: worst relation {max(st['rels']):.1e}  ||PC R^-1||_F median {np.median(st['amp']):.0f} max {max(st['amp']):.0f}")
    for seed in range(40):
        A, v1, kw, kind = case(seed)
        if A.dtype.kind == "c":
            continue
        st = run(src, A.toarray(), v1, kw, 20)["stats"]
        a = st.get("amp", [0.0])
        print(f"seed {seed:2d} {kind:9s} n={A.shape[0]:4d} {kw['mindim']}/{kw['maxdim']}: cycles {len(st['rels']):2d} worst relation {max(st['rels']):.1e}  ||PC R^-1||_F median {np.median(a):.0f} max {max(a):.0f}")


def watch():
    A, v1, kw, _ = case(17)
    src = SRC.replace(REL, '''_E = A @ V[:, :maxdim] - V @ H; _hn = np.linalg.norm(H); _r = float(np.linalg.norm(_E) / _hn)
            worst["rel"] = max(worst["rel"], _r); stats.setdefault("rels", []).append(_r)
            _w = float(np.linalg.norm(_E[:, max(k - 2, 0)]) / _hn) * np.sqrt(maxdim)
            WATCH(stats, _w, tol)''')
    src = src.replace(SLOOP, "        s = min(s_max, stats.get('s_cap', s_max), to - j + 1)\n        k = j\n")

    def make(enabled):
        def w(stats, x, tol):
            last, stats["w_last"] = stats.get("w_last"), x
            cap = stats.get("s_cap", 20)
            if enabled and last and x > 2e-14 and x / last > 2.0 and x * (x / last) > max(1e-11, 0.03 * tol) and cap > 2:
                stats["s_cap"] = max(2, cap // 2)
                stats.setdefault("log", []).append((len(stats["rels"]), "%.1e" % x, "x%.0f" % (x / last), stats["s_cap"]))
        return w
    for en in (False, True):
        st = run(src, A.toarray(), v1, kw, 20, restarts=40, WATCH=make(en))["stats"]
        print(f"growth-extrapolating watch {'ON ' if en else 'off'}: worst relation {max(st['rels']):.1e}, final {st['rels'][-1]:.1e}, block-size changes (cycle, measured, growth, new cap): {st.get('log')}")


def truncate():
    src = SRC.replace(REL, REL)
    src = src.replace('''        if not (ratio > pivot_min):
            raise BlockBail(j0, f"pivot ratio {ratio:.2e} at block column {i} (k = {k})")''', '''        if gref is not None and not (d > canc_min * gref[i]):
            ratio = ratios[-1] = -abs(d / gref[i])
        if not (ratio > pivot_min):
            if trunc_min is not None and i >= trunc_min:
                return R[:i, :i], min(ratios[:i]), i
            raise BlockBail(j0, f"pivot ratio {ratio:.2e} at block column {i} (k = {k})")''')
    src = src.replace("def chol_upper(G, pivot_min, k, j0):", "def chol_upper(G, pivot_min, k, j0, trunc_min=None, gref=None, canc_min=0.0):")
    src = src.replace("    return R, min(ratios)\n", "    return (R, min(ratios)) if trunc_min is None else (R, min(ratios), s)\n", 1)
    src = src.replace("        R1, piv1 = chol_upper(GZ - P.conj().T @ P, pivot_min, k, j)", '''        if TRUNC:
            R1, piv1, ngood = chol_upper(GZ - P.conj().T @ P, pivot_min, k, j, trunc_min=2, gref=np.real(np.diag(GZ)), canc_min=CANC)
            if ngood < s:
                stats["truncs"] = stats.get("truncs", 0) + 1
                stats["wasted"] = stats.get("wasted", 0) + (s - ngood)
                s = ngood
                Z = Z[:, :s]; Praw = Praw[:, :s]; GZ = GZ[:s, :s]; P = P[:, :s]; th = th[:s]; sig = sig[:s]
                stats["s_next"] = max(2, s)
        else:
            R1, piv1 = chol_upper(GZ - P.conj().T @ P, pivot_min, k, j)''')
    src = src.replace(SLOOP, "        s = min(s_max, stats.get('s_next', s_max), to - j + 1)\n        k = j\n")
    src = src.replace("    except BlockBail as b:\n", "    except BlockBail as b:\n        stats.setdefault('why', []).append(str(b)[:60])\n")
    from oracle.matrices import hashed_nonsymmetric
    rng = np.random.default_rng(3)
    n = 400
    M = rng.standard_normal((n, n)) / np.sqrt(n)
    M[0, 0] = 50.0
    pl = [(5.0, 3.0), (4.0, -2.5), (-6.0, 1.0), (3.5, 3.5), (-4.5, 2.0)]
    B = hashed_nonsymmetric(3000, seed=7, planted=pl).toarray()
    for name, A, kw in (("dense disc + outlier (test/partial_schur.jl:122-138) n=400 nev 5 LM 10/20", M, dict(nev=5, which="LM", tol=1e-10, mindim=10, maxdim=20, restarts=60)),
                        ("hashed 3000 with five planted pairs nev 10 LM 10/20", B, dict(nev=10, which="LM", tol=1e-10, mindim=10, maxdim=20, restarts=60))):
        v1 = rng.standard_normal(A.shape[0])
        for trunc, canc in ((False, 0.0), (True, 0.0), (True, 1e-7)):
            out = run(src, A, v1, kw, 20, TRUNC=trunc, CANC=canc)
            st = out["stats"]
            print(f"{name}: truncate {trunc} cancellation limit {canc:g}: products {out['prods']}, converged {len(out['eig'])}, blocks {st.get('blocks', 0)}, abandoned {st.get('bails', 0)}, "
                  f"truncated {st.get('truncs', 0)} (wasted products {st.get('wasted', 0)}), worst relation {out['worst']['rel']:.1e}; reasons {st.get('why')}")


if __name__ == "__main__":
    {"segments": segments, "amp": amp, "watch": watch, "truncate": truncate, "repair": repair}[sys.argv[1] if len(sys.argv) > 1 else "segments"]()
