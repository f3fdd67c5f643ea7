"""``opty_sincos`` (opty_amd/csrc/opty_device.h): a Python model of the fast
path -- same constants, parsed from the header; every FMA emulated exactly
with rational arithmetic -- against 50-digit mpmath values (CPU), and the
device function itself against NumPy over a sweep that includes the
out-of-line library path (GPU)."""
import math
import os
import re
from fractions import Fraction

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
HEADER = os.path.join(REPO, 'opty_amd', 'csrc', 'opty_device.h')


def _constants():
    with open(HEADER) as f:
        text = f.read()
    body = text[text.index('void opty_sincos(double x'):]
    body = body[:body.index('\n}\n')]
    nums = [float(x) for x in re.findall(
        r'(?<![\w.])-?\d\.\d{10,}e[-+]\d+', body)]
    limit = float(re.search(r'<= (\d+\.\d)\)', body).group(1))
    return limit, nums


def fma(a, b, c):
    return float(Fraction(a)*Fraction(b) + Fraction(c))


def model_sincos(x):
    limit, k = _constants()
    assert len(k) == 16 and abs(x) <= limit
    two_over_pi, hi, mid, lo = k[:4]
    S = k[4:10]             # S6, S5, S4, S3, S2, S1 in evaluation order
    C = k[10:16]            # C6 ... C1
    kk = float(round(x*two_over_pi))         # rint: half to even
    r = fma(-kk, hi, x)
    r = fma(-kk, mid, r)
    r = fma(-kk, lo, r)
    z = r*r
    ps = z*S[0] + S[1]       # multiply, add (as the device code)
    for c in S[2:]:
        ps = fma(z, ps, c)
    sr = fma(z*r, ps, r)
    pc = z*C[0] + C[1]
    for c in C[2:]:
        pc = fma(z, pc, c)
    hz = 0.5*z
    w = 1.0 - hz
    cr = w + (((1.0 - w) - hz) + (z*z)*pc)
    n = int(kk)
    a, b = (cr, sr) if n & 1 else (sr, cr)
    return (-a if n & 2 else a), (-b if (n + 1) & 2 else b), kk


def test_constants_are_what_they_claim():
    import mpmath
    mpmath.mp.dps = 60
    _, k = _constants()
    two_over_pi, hi, mid, lo = k[:4]
    half_pi = mpmath.pi/2
    assert hi == float(half_pi)
    assert mid == float(half_pi - hi)
    assert lo == float(half_pi - hi - mid)
    assert two_over_pi == float(2/mpmath.pi)
    # leading Taylor coefficients (the minimax ones sit next to them)
    assert abs(k[9] + 1.0/6) < 1e-15 and abs(k[15] - 1.0/24) < 1e-15


def test_fast_path_model_against_mpmath():
    import mpmath
    mpmath.mp.dps = 60
    rng = np.random.default_rng(7)
    xs = list(rng.uniform(-10, 10, 600)) + list(rng.uniform(-1, 1, 300))
    xs += [0.0, 1e-300, -1e-30, 2.0**-27, 0.78539816339744828,
           0.78539816339744839, 1048576.0, -1048576.0]
    half_pi = float(mpmath.pi/2)
    for kk in list(rng.integers(-40, 40, 150)) + \
            list(rng.integers(-600000, 600000, 150)):
        base = float(kk)*half_pi
        xs += [base, np.nextafter(base, np.inf), base + 1e-9,
               base*(1 + 3e-16) - 1e-13]
    xs += list(rng.uniform(-1048576, 1048576, 300))
    worst = 0.0
    for x in xs:
        x = float(x)
        if abs(x) > 1048576.0:
            continue
        s, c, kk = model_sincos(x)
        for got, ref in ((s, mpmath.sin(mpmath.mpf(x))),
                         (c, mpmath.cos(mpmath.mpf(x)))):
            err = abs(mpmath.mpf(got) - ref)
            # 1.25 ulp of the result, or -- next to a zero crossing
            # far from the origin -- what 53 bits of k*(pi/2 - hi) leave
            tol = 2.0**-53*max(2.5*abs(float(ref)), abs(kk)*6.2e-17)
            assert err <= tol, (x, got, float(ref), float(err), tol)
            if float(ref) != 0.0:
                worst = max(worst, float(err/abs(ref)))
    assert worst < 1e-10            # the parity bar, with a wide margin


@pytest.mark.gpu
def test_device_sincos_against_numpy():
    """sin / cos / both of one argument through generated matrix kernels:
    fast path, slow (library) path beyond 2^20, NaN, Inf."""
    import sympy as sm
    import opty_amd
    from opty_amd.codegen.emit_hip import EmitOptions
    a = sm.symbols('a')
    opts = dict(emit_options=EmitOptions(fast_trig=1))
    f = opty_amd.ufuncify_matrix((a,), sm.Matrix([[sm.sin(a), sm.cos(a)],
                                                  [sm.sin(a), 2*sm.cos(a)]]),
                                 **opts)
    f_sin = opty_amd.ufuncify_matrix((a,), sm.Matrix([[sm.sin(a)]]), **opts)
    f_cos = opty_amd.ufuncify_matrix((a,), sm.Matrix([[sm.cos(a)]]), **opts)
    assert 'opty_sincos(' in f.source and 'opty_sin(' in f_sin.source \
        and 'opty_cos(' in f_cos.source
    rng = np.random.default_rng(11)
    half_pi = np.pi/2
    k = rng.integers(-600000, 600000, 20000).astype(float)
    x = np.concatenate([
        rng.uniform(-10, 10, 60000), rng.uniform(-1, 1, 20000),
        k*half_pi + rng.uniform(-1e-6, 1e-6, k.size),
        rng.uniform(-1048576, 1048576, 20000),
        rng.uniform(-1e9, 1e9, 5000), [1e300, -1e22, 2.0**20, -2.0**20,
                                       np.nextafter(2.0**20, np.inf), 0.0,
                                       5e-324, np.nan, np.inf, -np.inf]])
    n = len(x)
    out = f(np.empty((n, 4)), x)
    s1 = f_sin(np.empty((n, 1)), x)[:, 0, 0]
    c1 = f_cos(np.empty((n, 1)), x)[:, 0, 0]
    with np.errstate(invalid='ignore'):
        s_ref, c_ref = np.sin(x), np.cos(x)
    fin = np.isfinite(x)
    assert np.isnan(out[~fin]).all() and np.isnan(s1[~fin]).all()
    kk = np.abs(np.rint(x[fin]*2/np.pi))
    for got, ref in ((out[fin, 0, 0], s_ref[fin]), (out[fin, 0, 1], c_ref[fin]),
                     (s1[fin], s_ref[fin]), (c1[fin], c_ref[fin]),
                     (out[fin, 1, 1]/2, c_ref[fin])):
        tol = 2.0**-53*np.maximum(4.0*np.abs(ref),
                                  np.minimum(kk, 2.0**20)*6.2e-17)
        err = np.abs(got - ref)
        bad = ~(err <= tol)
        assert not bad.any(), (x[fin][bad][:5], got[bad][:5], ref[bad][:5])
    np.testing.assert_array_equal(out[:, 0, 0], out[:, 1, 0])
