"""The t-test's tail as the DEVICE computes it (csrc/wt_core.h wt_tdist_2Q_fast: one asymptotic series for
lgamma(a + 1/2) - lgamma(a), the continued fraction by a division-free forward recurrence) against the form the oracle and
the emulator use (modified Lentz + lgamma, oracle/wt_oracle.c) and against scipy, over the (t, nu) plane.  Compiled for the
host from the same source (tests/emu); the device differs only in its libm."""
import ctypes as C
import math

import numpy as np
import pytest

from emu import build as emu_build


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(emu_build.build())
    for f in (L.wtemu_tdist_2q_fast, L.wtemu_tdist_2q):
        f.restype = C.c_double
        f.argtypes = [C.c_double, C.c_double]
    L.wtemu_lgamma_half_diff.restype = C.c_double
    L.wtemu_lgamma_half_diff.argtypes = [C.c_double]
    return L


def test_lgamma_half_difference(lib):
    for a in list(np.geomspace(1e-3, 1e5, 400)) + [0.5, 1.0, 1.5, 15.999, 16.0, 16.5, 49.0, 32767.0]:
        want = math.lgamma(a + 0.5) - math.lgamma(a)
        got = lib.wtemu_lgamma_half_diff(float(a))
        assert abs(got - want) <= 4e-15 * max(1.0, abs(want)) + 2e-15 * abs(math.lgamma(a)), (a, got, want)


def test_fast_tail_equals_the_oracle_form(lib):
    rng = np.random.default_rng(11)
    ts = np.concatenate([[0.0, 1e-300, 1e-8, 1e-3, 0.5, 1.0, 2.0, 5.0, 10.0, 37.0, 1e3, 1e8], rng.gamma(1.0, 2.0, 3000), rng.uniform(0, 60, 1000)])
    nus = np.concatenate([[1e-3, 0.5, 1.0, 2.0, 2.5, 3.0, 31.9, 32.0, 32.1, 98.0, 1000.0, 65534.0], rng.uniform(1.0, 200.0, 3000), np.geomspace(0.1, 6e4, 1000)])
    # both forms compute a * log(x) and (the oracle's) lgamma(a + 1/2) - lgamma(a) of two numbers ~ a ln a: their common
    # rounding grows with a = nu / 2 (at nu = 65534 the oracle's form is 1.9e-11 off the true value, this one 4e-12: mpmath)
    # (both forms take y = 1 - x = t^2 / (nu + t^2) as computed, not 1 - x of the rounded x: round 6 -- before, the oracle's form
    #  gave 0.99999999431700 for t = 1e-8, nu = 1/2, true 0.99999999460647)
    def tol(nu, b, t=1.0):
        # ... and below t = 3 (nu <= 2000) the device's form takes the fraction in y = 1 - x for the sake of the wavefront's slowest
        # lane: its result 1 - r carries the front factor's a * 1e-16 as an ABSOLUTE error
        wide = 3e-16 * (nu / 2 + 8.0) if t < 3.0 and nu <= 2000.0 else 0.0
        return (1e-12 + 1e-15 * nu * math.log(nu + 2.0)) * abs(b) + 1e-300 + wide
    for t in ts[:12]:
        for nu in nus[:12]:
            a, b = lib.wtemu_tdist_2q_fast(float(t), float(nu)), lib.wtemu_tdist_2q(float(t), float(nu))
            assert abs(a - b) <= tol(nu, b, t), (t, nu, a, b)
    for t, nu in zip(ts[12:], nus[12:]):
        a, b = lib.wtemu_tdist_2q_fast(float(t), float(nu)), lib.wtemu_tdist_2q(float(t), float(nu))
        assert abs(a - b) <= tol(nu, b, t), (t, nu, a, b)
    # the special values of wt_tdist_Q
    assert math.isnan(lib.wtemu_tdist_2q_fast(float("nan"), 5.0)) and math.isnan(lib.wtemu_tdist_2q_fast(1.0, float("nan")))
    assert math.isnan(lib.wtemu_tdist_2q_fast(1.0, 0.0)) and math.isnan(lib.wtemu_tdist_2q_fast(1.0, -3.0))
    assert lib.wtemu_tdist_2q_fast(float("inf"), 7.0) == 0.0
    assert lib.wtemu_tdist_2q_fast(0.0, 7.0) == 1.0


def test_fast_tail_against_scipy(lib):
    st = pytest.importorskip("scipy.stats")
    rng = np.random.default_rng(5)
    for t, nu in zip(rng.uniform(0, 40, 2000), rng.uniform(2.0, 200.0, 2000)):
        want = 2 * st.t.sf(t, nu)
        got = lib.wtemu_tdist_2q_fast(float(t), float(nu))
        assert abs(got - want) <= 1e-10 * want + 1e-300, (t, nu, got, want)


def test_fast_tail_against_mpmath(lib):
    """The true value (40 digits): the device's form is within 1e-12 + 1e-15 nu of it (a * log(x) with a = nu / 2 is what is
    left: x = nu / (nu + t^2) is itself rounded) -- closer than the oracle's form where nu is large."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    rng = np.random.default_rng(9)
    ts = np.concatenate([rng.gamma(1.0, 2.0, 300), rng.uniform(0, 40, 100), [0.5, 0.5]])
    nus = np.concatenate([rng.uniform(2.0, 200.0, 300), np.geomspace(1.0, 6e4, 100), [65534.0, 6665.027486277304]])
    for t, nu in zip(ts, nus):
        T, NU = mp.mpf(float(t)), mp.mpf(float(nu))
        want = float(mp.betainc(NU / 2, mp.mpf("0.5"), 0, NU / (NU + T * T), regularized=True))
        got = lib.wtemu_tdist_2q_fast(float(t), float(nu))
        wide = 3e-16 * (nu / 2 + 8.0) if t < 3.0 and nu <= 2000.0 else 0.0        # (see tol() above)
        assert abs(got - want) <= (1e-12 + 1e-15 * nu) * want + 1e-300 + wide, (t, nu, got, want)
