"""The kernels' arithmetic against values that do not come from libldpc_amd/csrc/detmath.h.

Tier 1 of the parity argument compares the HIP kernels with an oracle mode that includes the same
header; an error in dm_exp / dm_log / dm_ratio_* / the check-node forms would be invisible there.  These tests close
that: every function is evaluated on the device (ldpc_hip_selftest_math) and held against
  * the committed fixture tests/golden/math_ref.npz — glibc libm / x87 long double values (make_math.py),
  * the same reference computed live on >= 2^20 fresh points, and
  * detmath.h compiled for the host, bit for bit (the claim "same source, same bits" behind tier 1).
Tolerances are stated per function in TOL below, in units of the reference value's ulp plus an absolute floor.
"""
import os

import numpy as np
import pytest

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "math_ref.npz")

# |device - reference| <= ulps * ulp(reference) + floor(a, b)
#   exp family / log: both sides are < 1 ulp routines                                   -> 2 ulp
#   boxplus: sign*min + log term: an ulp of min(|x|,|y|) from the sum; the log term's argument (1+e^-s)/(1+e^-d) is
#            rounded three times on BOTH sides (1.1e-16 each, relative to a value in [0.5, 2]), measured 4.4e-16 apart
#   ratio forms: three roundings (numerator, denominator, quotient) against the exact value -> 3 ulp
#   check nodes in ratio form: D-2 steps of that, measured 5 ulp at D = 8                -> 8 ulp
#   shared-reciprocal forms (degree 3, 4): a product of up to four factors and a reciprocal in place of a quotient -> 8 ulp
#   check nodes in the LLR domain: results up to |L| = 600, 1 ulp(600) = 1.1e-13 per step  -> 4 ulp of the largest input
TOL = {"exp": (2, 0.0), "log": (2, 1e-300), "boxplus": (2, 6e-16), "ratio_div": (0, 0.0), "ratio_rho": (3, 0.0),
       "ratio_lambda": (3, 0.0), "e_combine": (3, 0.0), "exp_clamped": (2, 0.0), "boxplus_exp": (2, 0.0),
       "boxplus_log": (0, 3.5e-16), "cn_ratio3": (8, 0.0), "cn_ratio4": (8, 0.0), "cn_ratio5": (8, 0.0), "cn_ratio6": (8, 0.0),
       "cn_ratio8": (8, 0.0), "cn_llr4": (0, 0.0), "cn_llr6": (0, 0.0), "cn_ratio3s": (8, 0.0), "cn_ratio4s": (8, 0.0),
       "cn_ratio6s": (10, 0.0)}


def check(fn, a, b, got, ref):
    ulps, floor = TOL[fn]
    with np.errstate(all="ignore"):
        fin = np.isfinite(ref)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), fn
        assert np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)]), fn  # infinities as libm returns them
        tol = ulps * np.spacing(np.abs(ref)) + floor
        if fn == "boxplus":
            tol = tol + np.spacing(np.minimum(np.abs(a), np.abs(b)))
        if fn.startswith("cn_llr"):
            tol = 4 * np.spacing(np.abs(a).max(axis=1, keepdims=True)) + 2e-15
        bad = fin & ~(np.abs(got - ref) <= tol)
    assert not bad.any(), (fn, int(bad.sum()), a[bad][:3], got[bad][:3], ref[bad][:3])


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("fn", orc.MATH_FNS)
def test_fixture_is_what_libm_says_here(gold, fn):
    """The committed values against the same computation on this machine (glibc picks exp/log variants per CPU: they
    may differ in the last bit), and detmath.h on the host within the stated tolerance of them."""
    a, b, ref = gold[f"{fn}/a"], gold[f"{fn}/b"] if f"{fn}/b" in gold else None, gold[f"{fn}/ref"]
    with np.errstate(all="ignore"):
        live = orc.math_eval(fn, a, b)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(live), fin)
        assert (np.abs(live[fin] - ref[fin]) <= 4 * np.spacing(np.abs(ref[fin])) + 4e-16).all()
    check(fn, a, b, orc.math_eval(fn, a, b, det=True), ref)


@pytest.mark.parametrize("fn", ["boxplus", "cn_llr4", "cn_llr6"])
def test_host_header_vs_libm_on_live_points(fn):
    """detmath.h compiled for the host against libm / the libm box-plus chain on fresh points — saturated check nodes
    (magnitudes to 1e13, the edges of the rule: smallest input around 40, spread around 600) and saturated box-plus
    operands included (orc.math_points).  The device side of the same comparison needs a GPU (below)."""
    n = 1 << 15 if fn.startswith("cn_") else 1 << 18
    a, b = orc.math_points(fn, n, seed=11)
    with np.errstate(all="ignore"):
        check(fn, a, b, orc.math_eval(fn, a, b, det=True), orc.math_eval(fn, a, b))
    if fn.startswith("cn_"):
        mu, amax = np.abs(a).min(axis=1), np.abs(a).max(axis=1)
        sat = (mu >= 40) & (amax - mu <= 600)
        assert 0.2 < sat.mean() < 0.7 and ((amax > 600) & ~sat).any() and ((amax <= 600) & ~sat).any()  # all three forms taken


@pytest.fixture(scope="module")
def dec():
    import libldpc_amd
    return libldpc_amd.HipDecoder(orc.H_TXT)


@pytest.mark.gpu
@pytest.mark.parametrize("fn", orc.MATH_FNS)
def test_device_arithmetic_vs_fixture(dec, gold, fn):
    a, b, ref = gold[f"{fn}/a"], gold[f"{fn}/b"] if f"{fn}/b" in gold else None, gold[f"{fn}/ref"]
    check(fn, a, b, dec.selftest_math(fn, a, b), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("fn", orc.MATH_FNS)
def test_device_arithmetic_vs_live_libm_and_host_header(dec, fn):
    """2^20 fresh points per scalar function (2^17 rows per check-node function): within tolerance of libm / long
    double, and bit-identical to detmath.h compiled for the host."""
    n = 1 << 17 if fn.startswith("cn_") else 1 << 20
    a, b = orc.math_points(fn, n, seed=7)
    got = dec.selftest_math(fn, a, b)
    with np.errstate(all="ignore"):
        check(fn, a, b, got, orc.math_eval(fn, a, b))
        host = orc.math_eval(fn, a, b, det=True)
    same = (got == host) | (np.isnan(got) & np.isnan(host))
    assert same.all(), (fn, int((~same).sum()), a[~same][:3] if a.ndim == 1 else a[~same.all(axis=1)][:1])
    assert np.array_equal(np.signbit(got), np.signbit(host))  # signed zeros included
