"""GPU parity tests: the HIP path (through the C ABI of libldpc.so) against the CPU oracle and the
committed reference fixtures.  Run on the MI355X box: python -m pytest tests -m gpu

Parity definition:
  * everything that involves no exp/log (min-sum decoding, BSC/BEC channels, iteration counts, syndrome,
    hard decisions, bit-error counts) is bit-exact against the reference fixtures;
  * sum-product and the AWGN noise use exp/log: against the oracle built with the SAME deterministic
    exp/log (ORC_MATH_DET) every double is bit-identical; against the reference (glibc libm) LLRs agree
    within 1e-5 (north_star tolerance) and hard decisions / iteration counts are identical on converged
    frames.
"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: "LLRs within 1e-5"


@pytest.fixture(scope="module")
def dec():
    import libldpc_amd
    d = libldpc_amd.HipDecoder(orc.H_TXT)
    assert d.lib.ldpc_hip_device_count() >= 1, "no GPU visible"
    return d


@pytest.fixture(scope="module")
def ocode():
    return orc.Code(orc.H_TXT)


# ------------------------------------------------------------------------------------------------
def test_device_mt19937_64_matches_stream(dec):
    ref = orc.mt64_stream(0, 800000)
    got = dec.mt64(0, 0, 800000)  # crosses three 262080-word chunks: exercises the jump-ahead states
    assert np.array_equal(ref, got)
    for seed, first, n in [(5489, 9999, 1), (7, 262079, 3), (123456789, 524160 - 5, 700)]:
        ref = orc.mt64_stream(seed, first + n)[first:]
        assert np.array_equal(ref, dec.mt64(seed, first, n)), (seed, first, n)
    # ISO C++ known answer: 10000th output of mt19937_64(5489)
    assert int(dec.mt64(5489, 9999, 1)[0]) == 9981545732273789042


def test_device_mt19937_64_far_seek(dec):
    first = 40_000_000  # ~150 chunks ahead: states come from 8 doubling rounds
    ref = orc.mt64_stream(42, first + 1000)[first:]
    assert np.array_equal(ref, dec.mt64(42, first, 1000))


# ------------------------------------------------------------------------------------------------
def test_minsum_decode_bit_exact_vs_reference(dec, golden_frames):
    name = "awgn_ms_m5"
    llr = golden_frames[f"{name}/llr_in"]
    r = dec.decode_batch(llr, decoding="BP_MS")
    assert np.array_equal(r["iters"], golden_frames[f"{name}/iters"])
    assert np.array_equal(r["hard"], golden_frames[f"{name}/hard"])
    assert np.array_equal(r["llr_out"], golden_frames[f"{name}/llr_out"])  # doubles, bit for bit


def test_minsum_fixed_iterations_bit_exact(dec, golden_frames):
    name = "awgn_ms_m4_noearly_i7"
    r = dec.decode_batch(golden_frames[f"{name}/llr_in"], early_term=False, iterations=7, decoding="BP_MS")
    assert np.array_equal(r["iters"], golden_frames[f"{name}/iters"])
    assert np.array_equal(r["hard"], golden_frames[f"{name}/hard"])
    assert np.array_equal(r["llr_out"], golden_frames[f"{name}/llr_out"])


def test_bp_decode_bit_exact_vs_det_oracle(dec, ocode, golden_frames):
    for name, early in (("awgn_bp_m4", True), ("awgn_bp_m6_noearly", False), ("awgn_bp_m4_skip1216", True)):
        llr = golden_frames[f"{name}/llr_in"]
        r = dec.decode_batch(llr, early_term=early)
        for f in range(llr.shape[0]):
            it, out, hard = ocode.decode(llr[f], early_term=early, math=orc.MATH_DET)
            assert r["iters"][f] == it, (name, f)
            assert np.array_equal(r["hard"][f], hard), (name, f)
            assert np.array_equal(r["llr_out"][f], out), (name, f)  # bit-identical doubles, also when not converged


def test_bp_decode_vs_reference_tolerance(dec, golden_frames):
    name = "awgn_bp_m4"
    r = dec.decode_batch(golden_frames[f"{name}/llr_in"])
    assert np.array_equal(r["iters"], golden_frames[f"{name}/iters"])
    assert np.array_equal(r["hard"], golden_frames[f"{name}/hard"])
    assert np.max(np.abs(r["llr_out"] - golden_frames[f"{name}/llr_out"])) < TOL


# ------------------------------------------------------------------------------------------------
def _stream(dec, chan, x, seed, skip, count, **kw):
    dec.stream_begin(chan, seed, x)
    if skip:
        dec.stream_skip(skip)
    return dec.stream_decode(count, want=("iters", "bit_errors", "hard", "llr_out", "llr_in"), **kw)


def test_awgn_stream_bit_exact_vs_det_oracle(dec, ocode):
    for x, seed, skip, count, ms in [(-4.0, 0, 0, 24, False), (-5.0, 3, 5, 16, True), (-4.5, 9, 1216, 4, False)]:
        r = _stream(dec, "AWGN", x, seed, skip, count, decoding="BP_MS" if ms else "BP")
        o = ocode.run_frames("AWGN", x, seed=seed, skip=skip, count=count, min_sum=ms, math=orc.MATH_DET)
        for k in ("llr_in", "iters", "bit_errors", "hard", "llr_out"):
            assert np.array_equal(r[k], o[k]), (x, seed, k)
        assert dec.stream_raw_draws == o["raw_draws"]


def test_ratio_form_hands_back_escaped_frames(dec, ocode):
    """Sum-product with early termination runs in likelihood-ratio form; a frame whose values leave the box that
    form can represent is decoded again by the LLR-domain form.  At these points some frames of the
    batch take each route (checked on the oracle, which applies the same per-frame rule): still bit-exact."""
    mixed = 0
    for x, seed, count in [(1.0, 3, 64), (6.0, 3, 64), (10.0, 3, 64), (12.0, 5, 64)]:
        orc.ratio_stats(reset=True)
        o = ocode.run_frames("AWGN", x, seed=seed, count=count, math=orc.MATH_DET)
        done, escaped = orc.ratio_stats()
        assert done + escaped == count
        mixed += 0 < escaped < count
        r = _stream(dec, "AWGN", x, seed, 0, count, decoding="BP")
        for k in ("llr_in", "iters", "bit_errors", "hard", "llr_out"):
            assert np.array_equal(r[k], o[k]), (x, seed, k)
        # and without the LLR outputs (the instantiation the simulation loop uses)
        dec.stream_begin("AWGN", seed, x)
        r2 = dec.stream_decode(count, want=("iters", "bit_errors", "hard"), decoding="BP")
        for k in ("iters", "bit_errors", "hard"):
            assert np.array_equal(r2[k], o[k]), (x, seed, k)
    assert mixed >= 2


def test_awgn_stream_vs_reference(dec, golden_frames):
    for name, x, seed, skip, ms in [("awgn_bp_m4", -4.0, 0, 0, False), ("awgn_ms_m5", -5.0, 0, 0, True),
                                    ("awgn_bp_m4_skip1216", -4.0, 0, 1216, False)]:
        n = golden_frames[f"{name}/iters"].shape[0]
        r = _stream(dec, "AWGN", x, seed, skip, n, decoding="BP_MS" if ms else "BP")
        assert np.max(np.abs(r["llr_in"] - golden_frames[f"{name}/llr_in"])) < 1e-9
        conv = golden_frames[f"{name}/iters"] < 50
        assert np.array_equal(r["iters"][conv], golden_frames[f"{name}/iters"][conv])
        assert np.array_equal(r["hard"][conv], golden_frames[f"{name}/hard"][conv])
        assert np.max(np.abs(r["llr_out"][conv] - golden_frames[f"{name}/llr_out"][conv])) < TOL
        # frames the reference fails to decode also fail here (their LLRs are chaotic in the last ulp of
        # the noise sample, so their wrong bits are compared against the same-math oracle instead)
        assert np.array_equal(r["iters"], golden_frames[f"{name}/iters"])
        assert np.array_equal(r["bit_errors"] > 0, golden_frames[f"{name}/bit_errors"] > 0)


def test_bsc_stream_bit_exact_vs_reference(dec, golden_frames):
    # BSC: one draw per bit, LLR = +-delta: no transcendental on the device => the LLR-in is bit-exact
    # against the reference; min-sum decoding then is too.
    r = _stream(dec, "BSC", 0.24, 0, 0, 6, decoding="BP")
    assert np.array_equal(r["llr_in"], golden_frames["bsc_bp_024/llr_in"])
    assert np.array_equal(r["iters"], golden_frames["bsc_bp_024/iters"])
    assert np.array_equal(r["hard"], golden_frames["bsc_bp_024/hard"])
    assert np.max(np.abs(r["llr_out"] - golden_frames["bsc_bp_024/llr_out"])) < TOL


def test_bsc_minsum_vs_oracle(dec, ocode):
    r = _stream(dec, "BSC", 0.28, 1, 3, 8, decoding="BP_MS")
    o = ocode.run_frames("BSC", 0.28, seed=1, skip=3, count=8, min_sum=True)  # libm oracle == reference
    for k in ("llr_in", "iters", "bit_errors", "hard", "llr_out"):
        assert np.array_equal(r[k], o[k]), k


def test_counters_2000_frames_vs_reference(dec, golden_counters):
    """iteration count and bit errors of the first 2000 frames at -4 dB (two frame errors, first at 1216)."""
    dec.stream_begin("AWGN", 0, -4.0)
    r = dec.stream_decode(2000)
    ref_it, ref_be = golden_counters["awgn_bp_m4/iters"], golden_counters["awgn_bp_m4/bit_errors"]
    conv = ref_it < 50
    assert np.array_equal(r["iters"][conv], ref_it[conv])
    assert np.array_equal(r["bit_errors"][conv], ref_be[conv])
    assert np.array_equal(np.flatnonzero(r["bit_errors"]), np.flatnonzero(ref_be))


def test_minsum_counters_1000_frames(dec, golden_counters):
    dec.stream_begin("AWGN", 0, -4.5)
    r = dec.stream_decode(1000, decoding="BP_MS")
    ref_it, ref_be = golden_counters["awgn_ms_m45/iters"], golden_counters["awgn_ms_m45/bit_errors"]
    conv = ref_it < 50
    assert np.array_equal(r["iters"][conv], ref_it[conv])
    assert np.array_equal(r["bit_errors"][conv], ref_be[conv])
    assert np.array_equal(r["bit_errors"] > 0, ref_be > 0)
    # failing frames: same frames fail, and the bit-error totals agree to within a few percent
    assert abs(int(r["bit_errors"].sum()) - int(ref_be.sum())) < 0.05 * int(ref_be.sum())


def test_batch_split_is_invisible(dec):
    """Frame f is the f-th frame of the stream however the batches are cut."""
    dec.stream_begin("AWGN", 5, -4.2)
    a = dec.stream_decode(300, decoding="BP_MS")
    dec.stream_begin("AWGN", 5, -4.2)
    parts = [dec.stream_decode(n, decoding="BP_MS") for n in (1, 7, 92, 200)]
    assert np.array_equal(a["iters"], np.concatenate([p["iters"] for p in parts]))
    assert np.array_equal(a["bit_errors"], np.concatenate([p["bit_errors"] for p in parts]))


def test_division_sequence_is_the_ieee_division(dec):
    """The likelihood-ratio kernels divide with the hardware's division sequence minus operand scaling and
    special-case fix-up (detmath.h, dm_ratio_div): identical to the correctly rounded quotient on every one of
    2^27 operand pairs spread over the whole exponent range the form can produce."""
    assert dec.selftest_division(1 << 27, seed=7) == 0
    assert dec.selftest_division(1 << 20, seed=12345) == 0


def _oracle_chunk(args):
    """Worker (CPU only): frames [skip, skip+count) of the stream through the det-mode oracle."""
    x, seed, skip, count = args[:4]
    chan, ms = (args[4], args[5]) if len(args) > 4 else ("AWGN", False)
    o = orc.Code(orc.H_TXT).run_frames(chan, x, seed=seed, skip=skip, count=count, math=orc.MATH_DET, min_sum=ms,
                                       want_vectors=False)
    return o["iters"], o["bit_errors"]


def test_bulk_131072_frames_bit_exact_vs_det_oracle(dec):
    """Two full batches of the headline workload (131 072 consecutive frames at -4 dB: ≈180 of them fail, ≈100 run
    all 50 iterations): iteration count and bit-error count of every frame equal the oracle's.  The oracle side
    runs in 16 CPU processes over disjoint frame ranges (its RNG-only skip makes that cheap)."""
    import multiprocessing as mp
    n, parts = 131072, 16
    per = n // parts
    with mp.get_context("spawn").Pool(parts) as pool:
        res = pool.map(_oracle_chunk, [(-4.0, 0, k * per, per) for k in range(parts)])
    it = np.concatenate([r[0] for r in res])
    be = np.concatenate([r[1] for r in res])
    dec.stream_begin("AWGN", 0, -4.0)
    r = dec.stream_decode(n, want=("iters", "bit_errors"))
    assert np.array_equal(r["iters"], it)
    assert np.array_equal(r["bit_errors"], be)
    assert 100 <= int((be > 0).sum()) <= 300 and int((it >= 50).sum()) >= 30


def test_bulk_100000_frames_vs_reference(dec, golden_bulk):
    """100 000 consecutive frames against the reference's own per-frame counters (tests/golden/ref_bulk.npz, 132
    frame errors, 7 of them undetected): on every frame the reference converged on — right or wrong codeword — the
    iteration count and the bit-error count are identical, and exactly the same frames are in error."""
    ref_it, ref_be = golden_bulk["iters"], golden_bulk["bit_errors"]
    n = len(ref_it)
    dec.stream_begin("AWGN", 0, -4.0)
    r = dec.stream_decode(n, want=("iters", "bit_errors"))
    conv = ref_it < 50
    assert int(conv.sum()) == n - 125
    assert np.array_equal(r["iters"][conv], ref_it[conv])
    assert np.array_equal(r["bit_errors"][conv], ref_be[conv])
    assert np.array_equal(r["bit_errors"] > 0, ref_be > 0)
    assert np.array_equal(r["iters"] >= 50, ~conv)


@pytest.mark.parametrize("chan,x,ms,n", [("BSC", 0.24, False, 16384), ("AWGN", -4.5, True, 16384), ("BSC", 0.27, True, 8192)])
def test_bulk_other_modes_bit_exact_vs_det_oracle(dec, chan, x, ms, n):
    """The same bulk check for the BSC (sum-product in likelihood-ratio form on two-valued LLRs; ≈3 % of the frames
    fail at 0.24) and for min-sum: iteration count and bit-error count of every frame equal the oracle's."""
    import multiprocessing as mp
    parts = 16
    per = n // parts
    with mp.get_context("spawn").Pool(parts) as pool:
        res = pool.map(_oracle_chunk, [(x, 4, k * per, per, chan, ms) for k in range(parts)])
    it = np.concatenate([r[0] for r in res])
    be = np.concatenate([r[1] for r in res])
    dec.stream_begin(chan, 4, x)
    r = dec.stream_decode(n, want=("iters", "bit_errors"), decoding="BP_MS" if ms else "BP")
    assert np.array_equal(r["iters"], it)
    assert np.array_equal(r["bit_errors"], be)
    assert int((be > 0).sum()) > 10
