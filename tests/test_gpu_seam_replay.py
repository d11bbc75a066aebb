"""`-m gpu`: SEAM REPLAY -- whole solves driven through the C ABI the way the reference's unmodified code would drive a
`HipBasis` (the Julia glue itself cannot run here: no Julia runtime).

Level 1 (array-type seam, src/ArnoldiMethod.jl:81-92): the oracle's line-by-line restatement of `_partialschur`,
   `iterate_arnoldi!`, `orthogonalize!`, `reinitialize!` (oracle/arnoldi.py = src/run.jl:224-392, src/expansion.jl) runs
   UNCHANGED on a `DeviceBasis` (tests/seam_facade.py): every operation it applies to V becomes one verb of
   include/kschur.h -- ks_apply / ks_col_norm / ks_gemv_t / ks_gemv_n_sub / ks_col_div / ks_rotate / ks_col_copy
   (src/expansion.jl:81-108,121; src/run.jl:363-365,382-383).  H, Q and every decision stay in the caller's host code.
Level 2 (fused expansion): the same, with `iterate_arnoldi!` replaced by what KrylovSchurHIP.jl's method does:
   hand the caller-owned H over, vouch for the factorisation (ks_workspace_assert_arnoldi), ONE ks_iterate_arnoldi call,
   copy the new H columns back; the restart stays the caller's (oracle.smalldense) and rotates through the verbs.

Acceptance: identical mvproducts / nconverged / restarts to the pure-numpy oracle on the same inputs, Ritz values to
1e-9 relative, and the reference's two invariants on the downloaded result (test/partial_schur.jl:104-105)."""
import numpy as np
import pytest
import scipy.sparse as sp

from __graft_entry__ import import_package
from oracle import arnoldi as oa
from oracle.matrices import hashed_nonsymmetric, laplace1d, laplace3d
from seam_facade import DeviceBasis, DeviceOperator

pytestmark = pytest.mark.gpu
pkg = import_package()
EPS = np.finfo(np.float64).eps


def _shift_invert(n=400):
    import scipy.sparse.linalg as spla

    rng = np.random.default_rng(3)
    A = (laplace1d(n) + 1j * sp.diags(0.3 * rng.random(n))).tocsc().astype(np.complex128)
    sigma = 1.7 + 0.1j
    lu = spla.splu((A - sigma * sp.identity(n)).tocsc())

    class ShiftInvert:
        shape = (n, n)
        dtype = np.complex128

        def mul_(self, y, x):
            y[:] = lu.solve(x)

    return ShiftInvert()


def _configs():
    """BASELINE.json configs 1-4 at replay size (name, operator for the oracle, kwargs)."""
    planted = [(3.0, 0.0), (2.5, 1.0), (-2.8, 0.0), (2.2, 0.7), (-2.0, 1.5), (1.9, 0.0)]
    return [
        ("cfg1 tridiagonal n=100 nev=10 SR", laplace1d(100), dict(nev=10, which="SR", tol=1e-10)),
        ("cfg2-params Laplacian 10x11x12 nev=20 20/40 SR", laplace3d(10, 11, 12), dict(nev=20, which="SR", tol=1e-9, mindim=20, maxdim=40)),
        ("cfg3-flavour hashed nonsymmetric n=3000 nev=10 LM", hashed_nonsymmetric(3000, seed=11, planted=planted), dict(nev=10, which="LM", tol=1e-9)),
        ("cfg4-flavour ComplexF64 shift-invert callback n=400 nev=6 LM", _shift_invert(), dict(nev=6, which="LM", tol=1e-10)),
    ]


def _start_vector(n, cplx):
    v = oa.uniform_hash(20240917, np.arange(n))
    return (v + 1j * oa.uniform_hash(7, np.arange(n))) if cplx else v


def _device_side(A, maxdim):
    n = A.shape[0]
    dtype = oa.vtype(A)
    ctx = pkg.Context(0)
    ctx.profile_enable(True)
    op = pkg.as_operator(A, ctx)
    ws = pkg.ArnoldiWorkspace(n, maxdim, dtype, ctx=ctx)
    return ctx, op, ws, dtype


def _replay(A, kw, fused, vouch=True, sstep=None, v1=None, restarts=200):
    """The oracle's `partialschur` body (src/run.jl:100-129) on a DeviceBasis.  Returns (PartialSchur, History, calls, ws)."""
    n = A.shape[0]
    nev = kw["nev"]
    mindim = kw.get("mindim", min(max(10, nev), n))
    maxdim = kw.get("maxdim", min(max(20, 2 * nev), n))
    ctx, op, ws, dtype = _device_side(A, maxdim)
    if sstep is not None:
        ws.set_sstep(sstep)
    V = DeviceBasis(ws)
    H = np.zeros((maxdim + 1, maxdim), dtype=dtype, order="F")  # CALLER-owned, not the workspace's pinned H
    Q = np.zeros((maxdim, maxdim), dtype=dtype, order="F")
    ows = oa.ArnoldiWorkspace(V, H, V_tmp=V.alias(), Q=Q)  # ArnoldiWorkspace(V, H; V_tmp, Q)  src/ArnoldiMethod.jl:81-92
    v1 = (_start_vector(n, np.dtype(dtype).kind == "c") if v1 is None else np.asarray(v1)).astype(dtype)

    def _copy(v):
        v[:] = v1  # copyto!(v, v1)  src/run.jl:126

    dev_op = DeviceOperator(op, n, dtype)
    saved = oa.iterate_arnoldi
    stats = dict(steps=0, reorth=0, breakdowns=0, explicit_steps=0)

    def fused_iterate(A_, ows_, frm, to, st=None):
        # KrylovSchurHIP.jl: ArnoldiMethod.iterate_arnoldi!(A::HipOperator, arnoldi{<:HipBasis}, range)
        if frm > to:
            return ows_
        w = ows_.V.ws
        if frm > 1:
            w.H[:, : frm - 1] = ows_.H[:, : frm - 1]
        if vouch:
            w.assert_arnoldi(frm - 1)
        else:
            w.H[:, : frm - 1] = np.nan  # a library that read these columns would poison the new ones
        r = w.iterate_arnoldi(A_.op, frm, to)
        for key in stats:
            stats[key] += r[key]
        ows_.V.calls["iterate_arnoldi"] += 1
        for j in range(frm, to + 1):
            ows_.H[: j + 1, j - 1] = w.H[: j + 1, j - 1]
        return ows_

    try:
        if fused:
            oa.iterate_arnoldi = fused_iterate
        oa.reinitialize(ows, 0, _copy)  # src/run.jl:126
        dec, hist = oa._partialschur(dev_op, ows, mindim, maxdim, nev, kw["tol"], restarts, kw["which"], 0)
    finally:
        oa.iterate_arnoldi = saved
    return dec, hist, V.calls, ws, v1, stats, (ctx, op)


def _check(name, A, kw, dec, hist, v1):
    ref, rh = oa.partialschur(A, v1=v1, restarts=200, **kw)
    tag = f"{name}: replay {hist} (restarts {hist.restarts}) / oracle {rh} (restarts {rh.restarts})"
    assert rh.converged, tag
    assert hist.converged and hist.nconverged == rh.nconverged, tag
    assert hist.mvproducts == rh.mvproducts and hist.restarts == rh.restarts, tag
    scale = max(1.0, float(np.abs(ref.eigenvalues).max()))
    np.testing.assert_allclose(np.sort_complex(dec.eigenvalues), np.sort_complex(ref.eigenvalues), atol=1e-9 * scale, err_msg=tag)
    Qh, R = np.asarray(dec.Q), np.asarray(dec.R)
    AQ = np.column_stack([_apply(A, Qh[:, i]) for i in range(Qh.shape[1])])
    nb = max(1.0, float(np.abs(ref.eigenvalues).max()))
    assert np.linalg.norm(AQ - Qh @ R) <= 1e-7 * nb * np.sqrt(Qh.shape[1]), tag  # tol-level: locking, src/run.jl:206-208
    assert np.linalg.norm(Qh.conj().T @ Qh - np.eye(Qh.shape[1])) < 1e-12 * Qh.shape[1], tag


def _apply(A, x):
    y = np.empty_like(x)
    oa.apply_operator(A, y, x)
    return y


@pytest.mark.parametrize("idx", range(4))
def test_level1_verb_by_verb_replay_of_the_reference_driver(idx):
    name, A, kw = _configs()[idx]
    dec, hist, calls, ws, v1, _, keep = _replay(A, kw, fused=False)
    _check(name, A, kw, dec, hist, v1)
    # the solve really went through the verbs: one ks_apply per matrix-vector product, one rotation + one column copy
    # per restart plus the final rotation, and norms / projections for every orthogonalisation
    assert calls["apply"] == hist.mvproducts, (calls, hist)
    assert calls["col_copy"] == hist.restarts and calls["rotate"] >= hist.restarts, (calls, hist)
    assert calls["gemv_t"] >= hist.mvproducts and calls["gemv_n_sub"] >= hist.mvproducts and calls["norm"] >= 2 * hist.mvproducts
    assert "iterate_arnoldi" not in calls
    # verbs written by the caller: the library claims no provenance for this factorisation
    assert ws.provenance == -1


@pytest.mark.parametrize("idx", range(4))
def test_level2_fused_expansion_with_the_callers_restart_and_caller_owned_H(idx):
    name, A, kw = _configs()[idx]
    dec, hist, calls, ws, v1, stats, keep = _replay(A, kw, fused=True)
    _check(name, A, kw, dec, hist, v1)
    assert calls["iterate_arnoldi"] == hist.restarts + 1 and calls["apply"] == 0, calls
    assert stats["steps"] == hist.mvproducts, (stats, hist)
    assert calls["col_copy"] == hist.restarts and calls["rotate"] >= hist.restarts, (calls, hist)


def test_level2_takes_the_implicit_form_only_because_the_glue_vouches():
    """Same replay with and without ks_workspace_assert_arnoldi.  Without it the library must NOT lean on the caller's H
    or on the Arnoldi relation of columns it did not produce: it runs the explicit three-pass form (the second-pass update
    kernel `axpy` runs) and never reads the earlier columns of H -- they are poisoned with NaN here to prove it.  With it
    the two-pass form runs (no `axpy` launches).  Both give the oracle's trail."""
    name, A, kw = _configs()[1]
    for vouch in (True, False):
        dec, hist, calls, ws, v1, stats, (ctx, op) = _replay(A, kw, fused=True, vouch=vouch)
        _check(name, A, kw, dec, hist, v1)
        prof = ctx.profile_get()
        if vouch:
            # (no explicit second-pass kernel; after the first restart the steps go in BLOCKS -- the s-step expansion, whose
            # Newton shifts the library takes from the Hessenberg matrix it handed back, since this caller runs the restart
            # itself and never tells it the Ritz values -- so there are fewer projection launches than products)
            info = ws.sstep_info
            assert prof["axpy"]["count"] == 0 and 0 < prof["fused"]["count"] <= hist.mvproducts, prof
            assert info["blocks"] > 0, info
            # ... and every expansion after a restart of the caller's measured the relation it was about to lean on
            rel = ws.relation_info
            assert rel["probes"] >= hist.restarts - 1 > 0 and rel["breaks"] == 0, (rel, hist.restarts)
        else:
            assert prof["axpy"]["count"] == hist.mvproducts, prof


@pytest.mark.parametrize("seed", [1, 3, 5])
def test_level2_relation_probe_protects_a_caller_that_runs_the_restart_itself(seed):
    """Imaginary-part target on a real operator (the ill-posed selection of tests/test_gpu_factored_basis_stress.py): the
    reference's own restart (src/run.jl:298-339, :363-365 -- here the oracle's restatement, run by the CALLER on a device
    basis) cuts a 2 x 2 block of the real Schur form and the Arnoldi relation of the kept columns is off by ~1e-5 ||H|| from
    then on.  The library never sees that restart; blocks would amplify the error to O(1) residuals (round 4).  The
    expansion that follows `assert_arnoldi` measures the relation of the last kept column (ks_workspace_relation_probes)
    and switches the blocks off for the run.  Asserted: the probe ran and fired, no more than one expansion went in blocks,
    and the solve is the per-step solve: same trail as a workspace with the blocks off whenever nothing ran in blocks."""
    from test_gpu_factored_basis_stress import _ill_posed_case

    A, v1, kw = _ill_posed_case(seed)
    out = []
    for sstep in (10, 0):
        dec, hist, calls, ws, _, stats, keep = _replay(A, kw, fused=True, sstep=sstep, v1=v1, restarts=kw["restarts"])
        Q = np.asarray(dec.Q)
        orth = float(np.linalg.norm(Q.conj().T @ Q - np.eye(Q.shape[1]))) if Q.shape[1] else 0.0
        res = float(np.linalg.norm(A @ Q - Q @ np.asarray(dec.R))) / sp.linalg.norm(A) if Q.shape[1] else 0.0
        out.append((hist, np.sort_complex(np.asarray(dec.eigenvalues)), ws.relation_info, ws.sstep_info, orth, res))
    (h1, e1, rel1, info1, orth1, res1), (h0, e0, rel0, info0, orth0, res0) = out
    assert rel1["probes"] > 0 and rel1["breaks"] > 0 and rel1["worst_leak"] > 1e-9 and info1["s"] == 0, (rel1, info1)
    assert info1["blocks"] <= 2 and info0["blocks"] == 0 and rel0["probes"] == 0, (info1, info0, rel0)
    assert orth1 < 1e-11 * max(1, h1.nconverged) and orth0 < 1e-11 * max(1, h0.nconverged), (orth1, orth0)
    if info1["blocks"] == 0 and info1["abandoned"] == 0:
        assert (h1.mvproducts, h1.nconverged, h1.restarts) == (h0.mvproducts, h0.nconverged, h0.restarts), (h1, h0, info1, rel1)
        if h1.nconverged:
            assert np.abs(e1 - e0).max() <= 1e-9 * max(1.0, np.abs(e0).max())
    else:
        assert res1 <= 10 * res0 + 1e-6, (res1, res0)
