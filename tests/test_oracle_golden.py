"""Pin the CPU oracle (oracle/ldpc_oracle.c) against fixtures produced by the reference itself.

The reference has no decoder vectors of its own; tests/golden/*.npz were generated
by tests/golden/make_golden.py from the unmodified reference built into oracle/_ref/.
libm mode must reproduce them bit for bit (doubles compared with ==).
"""
import os

import numpy as np
import pytest

import orc


@pytest.fixture(scope="module")
def codes():
    return {False: orc.Code(orc.H_TXT), True: orc.Code(orc.H_TXT, orc.G_TXT)}


def _run(codes, case, **kw):
    g, ch, dec, it, early, seed, x, skip, cnt = case
    return codes[bool(g)].run_frames(ch, x, seed=seed, skip=skip, count=cnt, min_sum=(dec == "BP_MS"),
                                     early_term=bool(early), iters=it, bec_compat=True, **kw)


def test_code_parameters(codes):
    c = codes[True]
    # SURVEY §8 sizes of tests/code/h.txt and g.txt
    assert (c.nc, c.mc, c.nnz, c.nct, c.mct, c.kct, c.kc, c.max_degree) == (1152, 1024, 3456, 1024, 896, 128, 128, 15)
    assert (c.g_rows, c.g_cols, c.g_nnz) == (128, 1152, 18912)
    assert list(c.puncture) == list(range(256, 384)) and c.num_shorten == 0
    assert c.rank() == 1021


def test_frames_bit_exact(codes, golden_frames, golden_sim):
    for name, case in golden_sim["cases"].items():
        r = _run(codes, case)
        for k in ("iters", "bit_errors", "hard", "codeword", "llr_in", "llr_out"):
            ref = golden_frames[f"{name}/{k}"]
            got = r[k].astype(ref.dtype)
            assert np.array_equal(ref, got), f"{name}/{k}: {np.sum(ref != got)} mismatches"


def test_counters_bit_exact(codes, golden_counters, golden_sim):
    for name, case in golden_sim["counter_cases"].items():
        r = _run(codes, case, want_vectors=False)
        assert np.array_equal(golden_counters[f"{name}/iters"], r["iters"]), name
        assert np.array_equal(golden_counters[f"{name}/bit_errors"], r["bit_errors"]), name


def test_first_frame_error_is_frame_1216(golden_counters):
    # two frame errors in the first 2000 frames, the first one at (1-based) frame 1217
    be = golden_counters["awgn_bp_m4/bit_errors"]
    assert np.flatnonzero(be)[0] == 1216 and np.count_nonzero(be) == 2


def _fmt(x, fec, bec, frames, iters, nc):
    return "%f %.3e %.3e %d %.3e" % (x, fec / frames, bec / (frames * nc), frames, iters / frames)


def test_simulate_matches_reference_cli(codes, golden_sim):
    """orc_simulate == the reference CLI's result file (ldpcsim.cpp:97-263), 1 thread."""
    G = {"<G>": orc.G_TXT}
    for name, entry in golden_sim["cli"].items():
        a = [G.get(s, s) for s in entry["args"]]
        opt = {"-s": "0", "-i": "50", "--channel": "AWGN", "--decoding": "BP", "--max-frames": str(10**10),
               "--frame-error-count": "50", "-G": ""}
        i, early = 3, True
        while i < len(a):
            if a[i] == "--no-early-term":
                early, i = False, i + 1
            else:
                opt[a[i]] = a[i + 1]
                i += 2
        code = codes[bool(opt["-G"])]
        res = code.simulate(opt["--channel"], [float(a[0]), float(a[1]), float(a[2])], seed=int(opt["-s"]),
                            min_sum=opt["--decoding"] == "BP_MS", early_term=early, iters=int(opt["-i"]),
                            bec_compat=True, max_frames=int(opt["--max-frames"]),
                            min_fec=int(opt["--frame-error-count"]))
        lines = ["snr fer ber frames avg_iter"]
        xs = np.arange(len(res["totals"]))
        vals = []
        v = float(a[0])
        while v < float(a[1]):
            vals.append(v)
            v += float(a[2])
        if opt["--channel"] != "AWGN":
            vals = vals[::-1]
        for i in xs:
            frames, fec, bec, iters = (int(t) for t in res["totals"][i])
            # the reference only writes a line when a frame error happened, and the line reflects
            # the counters at the LAST frame error
            if fec == 0:
                lines.append("")
                continue
            lines.append("%f %.3e %.3e %d %.3e" % (vals[i], res["fer"][i], res["ber"][i], res["frames"][i],
                                                  res["avg_iter"][i]))
        assert lines == entry["lines"], name


def test_mt19937_64_known_answer():
    # ISO C++ [rand.predef]: the 10000th invocation of a default-constructed mt19937_64 is 9981545732273789042
    s = orc.mt64_stream(5489, 10000)
    assert int(s[-1]) == 9981545732273789042


def test_detmath_close_to_libm():
    rng = np.random.default_rng(0)
    L = orc.lib()
    for x in -40 * rng.random(2000):
        a, b = L.orc_exp(orc.MATH_DET, x), L.orc_exp(orc.MATH_LIBM, x)
        assert abs(a - b) <= 2 * np.spacing(b)
    for q in 0.5 + 1.5 * rng.random(2000):
        a, b = L.orc_log(orc.MATH_DET, q), L.orc_log(orc.MATH_LIBM, q)
        assert abs(a - b) <= 2 * np.spacing(abs(b)) + 1e-300


def test_det_mode_within_tolerance_of_libm(codes, golden_frames):
    """north_star tolerance: LLRs within 1e-5, hard decisions / iteration counts exact (converged frames)."""
    r = codes[False].run_frames("AWGN", -4.0, seed=0, count=8, math=orc.MATH_DET)
    assert np.array_equal(r["iters"], golden_frames["awgn_bp_m4/iters"])
    assert np.array_equal(r["hard"], golden_frames["awgn_bp_m4/hard"])
    assert np.max(np.abs(r["llr_in"] - golden_frames["awgn_bp_m4/llr_in"])) < 1e-12
    assert np.max(np.abs(r["llr_out"] - golden_frames["awgn_bp_m4/llr_out"])) < 1e-5


def test_8k_code_frames_bit_exact(h8k_file, golden_8k, golden_sim):
    """config 4: (3,6)-regular nc=8192 — oracle == reference on AWGN/BSC/BEC frames."""
    code = orc.Code(h8k_file)
    assert (code.nc, code.mc, code.nnz, code.max_degree) == (8192, 4096, 24576, 6)
    for name, (ch, dec, it, early, seed, x, skip, cnt) in golden_sim["h8k"]["cases"].items():
        r = code.run_frames(ch, x, seed=seed, skip=skip, count=cnt, min_sum=(dec == "BP_MS"), early_term=bool(early),
                            iters=it, bec_compat=True)
        for k in ("iters", "bit_errors", "hard", "llr_in", "llr_out"):
            ref = golden_8k[f"{name}/{k}"]
            assert np.array_equal(ref, r[k].astype(ref.dtype)), f"{name}/{k}"


def test_shortened_code_frames_bit_exact(hshort_file, golden_frames, golden_sim):
    """shortening (LLR 99999.9 / delta / known symbol) incl. two shortened bits on one check node"""
    code = orc.Code(hshort_file)
    # column 290 is punctured AND shortened: nct = 1152 - 128 - 4 counts it twice (ldpc.h:55)
    assert (code.nct, code.kct, code.num_shorten) == (1020, 124, 4)
    for name, (ch, dec, it, early, seed, x, skip, cnt) in golden_sim["short_cases"].items():
        r = code.run_frames(ch, x, seed=seed, skip=skip, count=cnt, min_sum=(dec == "BP_MS"), early_term=bool(early),
                            iters=it, bec_compat=True)
        for k in ("iters", "bit_errors", "hard", "llr_in", "llr_out", "codeword"):
            ref = golden_frames[f"short/{name}/{k}"]
            assert np.array_equal(ref, r[k].astype(ref.dtype)), f"{name}/{k}"


def _bulk_chunk(args):
    math, skip, count = args
    o = orc.Code(orc.H_TXT).run_frames("AWGN", -4.0, seed=0, skip=skip, count=count, math=math, want_vectors=False)
    return o["iters"], o["bit_errors"]


def test_bulk_reference_counters(golden_bulk):
    """First 40 000 frames of the headline workload against the reference's own per-frame counters
    (tests/golden/ref_bulk.npz): the libm-mode oracle equals them on every frame, failing ones included; the
    det-mode oracle (the arithmetic of the HIP kernels: likelihood-ratio form, deterministic exp/log) equals them on
    every frame the reference converged on and fails exactly the same frames."""
    import multiprocessing as mp
    n, parts = 40000, 8
    per = n // parts
    ref_it, ref_be = golden_bulk["iters"][:n], golden_bulk["bit_errors"][:n]
    with mp.get_context("fork").Pool(parts) as pool:
        libm = pool.map(_bulk_chunk, [(orc.MATH_LIBM, k * per, per) for k in range(parts)])
        det = pool.map(_bulk_chunk, [(orc.MATH_DET, k * per, per) for k in range(parts)])
    it = np.concatenate([r[0] for r in libm])
    be = np.concatenate([r[1] for r in libm])
    assert np.array_equal(it, ref_it) and np.array_equal(be, ref_be)
    it = np.concatenate([r[0] for r in det])
    be = np.concatenate([r[1] for r in det])
    conv = ref_it < 50
    assert np.array_equal(it[conv], ref_it[conv]) and np.array_equal(be[conv], ref_be[conv])
    assert np.array_equal(be > 0, ref_be > 0)
    # the frames the reference fails: the det-mode oracle's bit-error counts and hard decisions against the reference's
    # own (tests/golden/ref_bulk_fail.npz); measured: none differs
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_bulk_fail.npz"))
    frames = fx["frames"][fx["frames"] < n]
    ref_hard = np.unpackbits(fx["hard_packed"], axis=1)[:len(frames), :1152]
    code = orc.Code(orc.H_TXT)
    different = 0
    for f, h in zip(frames, ref_hard):
        o = code.run_frames("AWGN", -4.0, seed=0, skip=int(f), count=1, math=orc.MATH_DET)
        different += int(o["bit_errors"][0] != ref_be[f] or not np.array_equal(o["hard"][0], h))
    assert len(frames) > 40 and different <= 1, different


def test_shape_fixtures_bit_exact(tmp_path):
    """tests/golden/ref_shapes.npz (make_shapes.py: the reference on a weight-20 check-node code and on an irregular
    7936-column code): the libm-mode oracle reproduces iteration counts, bit errors, hard decisions and LLR-out doubles
    bit for bit; the regenerated code files have the recorded sha256."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_shapes
    fx = np.load(os.path.join(orc.GOLDEN, "ref_shapes.npz"))
    codes = {}
    for name in make_shapes.CODES:
        path = make_shapes.code_file(name, str(tmp_path))
        assert hashlib.sha256(open(path, "rb").read()).digest() == fx[f"sha256/{name}"].tobytes(), name
        codes[name] = orc.Code(path)
    for key, (code, ch, dec, it, early, seed, x, skip, cnt) in make_shapes.CASES.items():
        r = codes[code].run_frames(ch, x, seed=seed, skip=skip, count=cnt, min_sum=(dec == "BP_MS"), early_term=bool(early), iters=it)
        assert np.array_equal(r["iters"], fx[f"{key}/iters"]) and np.array_equal(r["bit_errors"], fx[f"{key}/bit_errors"]), key
        assert np.array_equal(np.packbits(r["hard"], axis=1), fx[f"{key}/hard_packed"]), key
        assert np.array_equal(r["llr_out"], fx[f"{key}/llr_out"]), key


def test_degree6_three_stage_rule_on_the_8k_code(h8k_file):
    """The det oracle's three stages for codes the LDS-resident decoder does not take (shared reciprocals of degree-6 check
    nodes, separately divided outputs, LLR domain — detmath.h, dm_cn6_shared): on inputs crafted to need the second and the
    third stage every stage's result is the reference arithmetic's (libm oracle, itself pinned by the reference's fixtures
    above) — same iteration count and decisions, LLRs within the parity bound 1e-5 relative."""
    code = orc.Code(h8k_file)
    rng = np.random.default_rng(8)
    kinds = ["odd", "even", "plain", "odd"]
    frames = [orc.craft_degree6_overflow(code, rng, odd=(k == "odd")) if k != "plain" else np.where(rng.random(code.nc) < 0.02, -1.5, 2.5)
              for k in kinds]
    orc.ratio_stats(reset=True)
    det = [code.decode(f, math=orc.MATH_DET) for f in frames]
    done, escaped = orc.ratio_stats()
    assert (orc.ratio_second(), done, escaped) == (3, 3, 1)
    for f, (it, out, hard) in zip(frames, det):
        it2, out2, hard2 = code.decode(f, math=orc.MATH_LIBM)
        assert it == it2 and np.array_equal(hard, hard2)
        assert np.allclose(out, out2, rtol=1e-5, atol=1e-9)
