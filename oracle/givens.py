"""ORACLE (test infrastructure, never shipped in the product path).

`givensAlgorithm(f, g) -> (c, s, r)` such that

        [  c        s ] [ f ]   [ r ]
        [ -conj(s)  c ] [ g ] = [ 0 ]

The reference calls `LinearAlgebra.givensAlgorithm` from the Julia standard
library (NOT under /root/reference; un-pinned, `Project.toml:7,13` says
`julia = "1.6"`).  Call sites: `src/schurfact.jl:58,66-67`,
`src/schursort.jl:224-237,260-268,288-289`, `src/restore_hessenberg.jl:91`.
Julia's routine is a port of LAPACK 3.x `dlartg` / `zlartg`; what follows
restates that published algorithm (sign convention included: for real input
`c >= 0` whenever |f| > |g|; for complex input `c` is real and non-negative
and `r` carries the phase of `f`).
"""
from __future__ import annotations

import math

import numpy as np

_SAFMIN = np.finfo(np.float64).tiny
_EPS = np.finfo(np.float64).eps
# safmn2 = 2^trunc(log(safmin/eps)/log(2)/2)
_SAFMN2 = 2.0 ** math.trunc(math.log(_SAFMIN / _EPS) / math.log(2.0) / 2.0)
_SAFMX2 = 1.0 / _SAFMN2


def givens_real(f: float, g: float):
    """LAPACK dlartg (3.x) as ported in Julia's `givensAlgorithm(::T, ::T) where T<:AbstractFloat`."""
    f = float(f)
    g = float(g)
    if g == 0.0:
        return 1.0, 0.0, f
    if f == 0.0:
        return 0.0, 1.0, g
    f1, g1 = f, g
    scale = max(abs(f1), abs(g1))
    if scale >= _SAFMX2:
        count = 0
        while True:
            count += 1
            f1 *= _SAFMN2
            g1 *= _SAFMN2
            scale = max(abs(f1), abs(g1))
            if scale < _SAFMX2:
                break
        r = math.sqrt(f1 * f1 + g1 * g1)
        cs = f1 / r
        sn = g1 / r
        for _ in range(count):
            r *= _SAFMX2
    elif scale <= _SAFMN2:
        count = 0
        while True:
            count += 1
            f1 *= _SAFMX2
            g1 *= _SAFMX2
            scale = max(abs(f1), abs(g1))
            if scale > _SAFMN2:
                break
        r = math.sqrt(f1 * f1 + g1 * g1)
        cs = f1 / r
        sn = g1 / r
        for _ in range(count):
            r *= _SAFMN2
    else:
        r = math.sqrt(f1 * f1 + g1 * g1)
        cs = f1 / r
        sn = g1 / r
    if abs(f) > abs(g) and cs < 0.0:
        cs, sn, r = -cs, -sn, -r
    return cs, sn, r


def _abs1(z: complex) -> float:
    return max(abs(z.real), abs(z.imag))


def _abssq(z: complex) -> float:
    return z.real * z.real + z.imag * z.imag


def givens_complex(f: complex, g: complex):
    """LAPACK zlartg (3.x) as ported in Julia's `givensAlgorithm(::Complex{T}, ::Complex{T})`.

    Returns (c real, s complex, r complex)."""
    f = complex(f)
    g = complex(g)
    scale = max(_abs1(f), _abs1(g))
    fs, gs = f, g
    count = 0
    if scale >= _SAFMX2:
        while True:
            count += 1
            fs *= _SAFMN2
            gs *= _SAFMN2
            scale *= _SAFMN2
            if scale < _SAFMX2:
                break
    elif scale <= _SAFMN2:
        if g == 0:
            return 1.0, 0j, f
        while True:
            count -= 1
            fs *= _SAFMX2
            gs *= _SAFMX2
            scale *= _SAFMX2
            if scale > _SAFMN2:
                break
    f2 = _abssq(fs)
    g2 = _abssq(gs)
    if f2 <= max(g2, 1.0) * _SAFMIN:
        # f is negligible next to g
        if f == 0:
            cs = 0.0
            r = complex(math.hypot(g.real, g.imag), 0.0)
            d = math.hypot(gs.real, gs.imag)
            sn = complex(gs.real / d, -gs.imag / d)
            return cs, sn, r
        f2s = math.hypot(fs.real, fs.imag)
        g2s = math.sqrt(g2)
        cs = f2s / g2s
        if _abs1(f) > 1.0:
            d = math.hypot(f.real, f.imag)
            ff = complex(f.real / d, f.imag / d)
        else:
            dr = _SAFMX2 * f.real
            di = _SAFMX2 * f.imag
            d = math.hypot(dr, di)
            ff = complex(dr / d, di / d)
        sn = ff * complex(gs.real / g2s, -gs.imag / g2s)
        r = cs * f + sn * g
    else:
        f2s = math.sqrt(1.0 + g2 / f2)
        r = complex(f2s * fs.real, f2s * fs.imag)
        cs = 1.0 / f2s
        d = f2 + g2
        sn = complex(r.real / d, r.imag / d)
        sn = sn * gs.conjugate()
        if count != 0:
            if count > 0:
                for _ in range(count):
                    r *= _SAFMX2
            else:
                for _ in range(-count):
                    r *= _SAFMN2
    return cs, sn, r


def givens(f, g, is_real: bool):
    if is_real:
        return givens_real(f, g)
    return givens_complex(f, g)
