"""Whole decodes on the odd code shapes held against the REFERENCE's arithmetic (round 3).

tests/test_gpu_random_codes.py compares those shapes with the det-mode oracle, which shares libldpc_amd/csrc/detmath.h
with the kernels.  Here the same kernels — wide check nodes through the scratch form, the generic totals-form register
kernel, saturated check nodes, hand-over — are compared with (a) fixtures the unmodified reference produced
(tests/golden/ref_shapes.npz, make_shapes.py) and (b) the libm-mode oracle, which tests/test_oracle_golden.py shows
equal to the reference bit for bit.  north_star's rule: iteration counts and hard decisions identical on every frame the
reference converged on, the same frames failing, LLR-out within 1e-5 (relative above |L| = 1)."""
import os
import sys
import zlib

import numpy as np
import pytest

import orc
from test_gpu_random_codes import CASES, make_code, make_code_by_degrees

pytestmark = pytest.mark.gpu
TOL = 1e-5  # north_star: LLRs within 1e-5


def assert_reference_parity(code, got, ref, early, iters, what):
    """got: the GPU's outputs, ref: the reference's (fixture or libm-mode oracle) for the same frames."""
    n = len(ref["iters"])
    ref_hard = ref["hard"]
    if early:
        converged = ref["iters"] < iters
    else:  # fixed iterations: a frame has converged when its decisions satisfy every check
        converged = np.array([not code.syndrome(ref_hard[f]).any() for f in range(n)])
    assert np.array_equal(got["iters"][converged], ref["iters"][converged].astype(got["iters"].dtype)), what
    assert np.array_equal(got["hard"][converged], ref_hard[converged]), what
    assert np.array_equal(got["bit_errors"][converged], ref["bit_errors"][converged].astype(got["bit_errors"].dtype)), what
    if early:  # the same frames fail to converge
        assert np.array_equal(got["iters"] < iters, converged), what
    else:
        assert np.array_equal(np.array([not code.syndrome(got["hard"][f]).any() for f in range(n)]), converged), what
    a, b = got["llr_out"][converged], ref["llr_out"][converged]
    fin = np.isfinite(b)
    assert np.array_equal(a[~fin], b[~fin]), what
    err = np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin]))
    assert err.size == 0 or err.max() <= TOL, (what, float(err.max()))
    return int(converged.sum())


def test_reference_fixtures_wide_and_irregular(tmp_path):
    """GPU vs tests/golden/ref_shapes.npz: the wide-node scratch form (row weight 20, memory-resident) and the generic
    totals-form register kernel (irregular 7936-column code), sum-product with and without early termination, min-sum,
    AWGN and BSC."""
    import hashlib
    import libldpc_amd
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_shapes
    fx = np.load(os.path.join(orc.GOLDEN, "ref_shapes.npz"))
    decs, codes = {}, {}
    for name in make_shapes.CODES:
        path = make_shapes.code_file(name, str(tmp_path))
        assert hashlib.sha256(open(path, "rb").read()).digest() == fx[f"sha256/{name}"].tobytes(), name
        decs[name], codes[name] = libldpc_amd.HipDecoder(path), orc.Code(path)
    assert decs["wide20"].residency == "memory" and decs["irr"].residency == "registers" and decs["irr"].register_form == "totals"
    n_conv = 0
    for key, (code, ch, dec, it, early, seed, x, skip, cnt) in make_shapes.CASES.items():
        d = decs[code]
        d.stream_begin(ch, seed, x)
        d.stream_skip(skip)
        got = d.stream_decode(cnt, early_term=bool(early), iterations=it, decoding=dec, want=("iters", "bit_errors", "hard", "llr_out"))
        ref = {"iters": fx[f"{key}/iters"], "bit_errors": fx[f"{key}/bit_errors"], "llr_out": fx[f"{key}/llr_out"],
               "hard": np.unpackbits(fx[f"{key}/hard_packed"], axis=1)[:, :d.nc]}
        n_conv += assert_reference_parity(codes[code], got, ref, bool(early), it, key)
        if dec == "BP_MS":  # no transcendental in the decoder itself (the channel's normals still carry the polar method's log):
            for k in ("iters", "bit_errors", "hard"):  # every frame, failing ones included
                assert np.array_equal(got[k], ref[k].astype(got[k].dtype)), (key, k)
            assert np.abs(got["llr_out"] - ref["llr_out"]).max() < 1e-6, key
    assert n_conv >= 20


@pytest.mark.parametrize("case", [c for c in CASES if c[0] != "bec_beyond_lds"], ids=[c[0] for c in CASES if c[0] != "bec_beyond_lds"])
def test_random_code_vs_reference_arithmetic(case, tmp_path):
    """Every code of test_random_code_bit_exact (all residencies, check nodes up to weight 70, puncturing, shortening,
    isolated and degree-1 variable nodes) against the libm-mode oracle = the reference's arithmetic."""
    import libldpc_amd
    name, nc, mc, pool, punct, short, skip, residency = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    degs = rng.choice(pool, size=mc)
    path = make_code(str(tmp_path / f"{name}.txt"), nc, mc, degs, rng, punct, short, skip)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    n_conv = 0
    for ch, x, ms, early, iters in (("AWGN", 3.0, False, True, 20), ("AWGN", 2.0, False, False, 4), ("AWGN", 9.0, False, True, 20),
                                    ("AWGN", 13.0, False, True, 20), ("BSC", 0.03, False, True, 20)):
        d.stream_begin(ch, 3, x)
        d.stream_skip(1)
        got = d.stream_decode(5, early_term=early, iterations=iters, decoding="BP", want=("iters", "bit_errors", "hard", "llr_out"))
        ref = code.run_frames(ch, x, seed=3, skip=1, count=5, early_term=early, iters=iters, math=orc.MATH_LIBM)
        n_conv += assert_reference_parity(code, got, ref, early, iters, (name, ch, x))
    assert n_conv > 0


@pytest.mark.parametrize("name,vn,cn", [
    ("lds_regular_3_6", [3] * 600, [6] * 300),
    ("lds_mixed", [3] * 500 + [4] * 100 + [6] * 50, [4] * 300 + [5] * 120 + [8] * 50),
    ("mem_wide_cn20", [3] * 800, [20] * 120),
    ("mem_wide_cn40_24", [3] * 800, [40] * 30 + [24] * 50),
])
def test_saturated_decodes_vs_reference_arithmetic(name, vn, cn, tmp_path):
    """test_saturated_check_nodes_in_a_whole_decode's runs (sum-product without early termination, LLRs beyond 1e6: the
    saturated check-node form, the hand-over) against the libm-mode oracle: decisions, iteration counts, and LLR-out to
    1e-5 relative — at magnitudes where the reference itself evaluates log((1+e^-a)/(1+e^-b)) with both exponentials
    underflowing."""
    import libldpc_amd
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    path = make_code_by_degrees(str(tmp_path / f"{name}.txt"), vn, cn, rng)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    for x, iters in ((10.0, 45), (5.0, 30)):
        d.stream_begin("AWGN", 9, x)
        got = d.stream_decode(6, early_term=False, iterations=iters, decoding="BP", want=("iters", "bit_errors", "hard", "llr_out"))
        ref = code.run_frames("AWGN", x, seed=9, skip=0, count=6, early_term=False, iters=iters, math=orc.MATH_LIBM)
        assert_reference_parity(code, got, ref, False, iters, (name, x))


def test_irregular_totals_form_vs_reference_arithmetic(tmp_path):
    """test_irregular_code_totals_form's runs against the libm-mode oracle."""
    import libldpc_amd
    rng = np.random.default_rng(11)
    path = make_code_by_degrees(str(tmp_path / "irr.txt"), [2] * 3008 + [3] * 4928, [5] * 560 + [6] * 3000, rng)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    for x, early, iters in ((3.0, True, 30), (2.2, True, 12), (3.0, False, 40), (2.6, True, 25)):
        d.stream_begin("AWGN", 5, x)
        got = d.stream_decode(6, early_term=early, iterations=iters, decoding="BP", want=("iters", "bit_errors", "hard", "llr_out"))
        ref = code.run_frames("AWGN", x, seed=5, skip=0, count=6, early_term=early, iters=iters, math=orc.MATH_LIBM)
        assert_reference_parity(code, got, ref, early, iters, x)
