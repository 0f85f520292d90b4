"""GPU vs. same-math oracle on randomly generated irregular codes: every decoder residency (LDS, registers,
memory), every CN-width instantiation, degree-2 check nodes, isolated and degree-1 variable nodes, puncturing
and shortening.  Everything is compared bit for bit (ORC_MATH_DET oracle)."""
import zlib

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
OUT = ("iters", "bit_errors", "hard", "llr_out", "llr_in", "codeword")


def make_code(path, nc, mc, cn_degs, rng, puncture=(), shorten=(), skip_cols=()):
    """Random bipartite graph: check i gets cn_degs[i] distinct columns (never one in skip_cols)."""
    cols_ok = np.array([c for c in range(nc) if c not in set(skip_cols)])
    lines = []
    if puncture:
        lines.append(f"puncture [{len(puncture)}]: " + " ".join(map(str, puncture)))
    if shorten:
        lines.append(f"shorten [{len(shorten)}]: " + " ".join(map(str, shorten)))
    used = np.zeros(nc, bool)
    for i in range(mc):
        cs = rng.choice(cols_ok, size=cn_degs[i], replace=False)
        used[cs] = True
        for c in sorted(cs):
            lines.append(f"{i} {c}")
    # make sure the highest column index appears so that nc is what we asked for
    if not used[nc - 1]:
        lines.append(f"{mc - 1} {nc - 1}") if f"{mc - 1} {nc - 1}" not in lines else None
    open(path, "w").write("\n".join(lines))
    return path


CASES = [
    # name, nc, mc, CN degree pool, puncture, shorten, skipped columns, expected residency
    ("lds_small_irregular", 300, 150, [2, 3, 4], (5, 6, 7), (), (), "lds"),
    ("lds_wide_cn", 600, 200, [5, 6, 7, 8], (), (10, 11), (3,), "lds"),
    ("mem_cn12", 900, 100, [9, 12, 16], (), (), (), "memory"),
    ("reg_tile_8x4", 9000, 6000, [3, 4], (1, 2, 3), (), (7,), "registers"),
    ("reg_tile_2x8", 3000, 2000, [8], (), (), (), "registers"),
    # high-rate codes: check nodes wider than the register tiles (the reference takes any row weight,
    # decoder.cpp:25-45): through the scratch array of the memory-resident decoder
    ("mem_cn20", 1200, 120, [20], (), (), (), "memory"),
    ("mem_cn32_mixed", 2000, 150, [17, 24, 32, 70, 6], (4, 5), (9,), (), "memory"),
    # a code whose erasure-decoder state (nnz + 2 nc bytes) exceeds 160 KB of LDS: BEC state in device memory
    ("bec_beyond_lds", 70000, 35000, [4], (), (), (), None),
]


def make_code_by_degrees(path, vn_degs, cn_degs, rng):
    """Random graph with prescribed column AND row degrees (socket matching, duplicate edges repaired by swaps)."""
    assert sum(vn_degs) == sum(cn_degs)
    cols = np.repeat(np.arange(len(vn_degs)), vn_degs)
    rows = np.repeat(np.arange(len(cn_degs)), cn_degs)
    rng.shuffle(cols)
    for _ in range(200):
        key = rows.astype(np.int64) * len(vn_degs) + cols
        _, first = np.unique(key, return_index=True)
        dup = np.setdiff1d(np.arange(key.size), first)
        if dup.size == 0:
            break
        other = rng.integers(0, key.size, dup.size)
        cols[dup], cols[other] = cols[other].copy(), cols[dup].copy()
    else:
        raise AssertionError("could not repair duplicate edges")
    order = np.lexsort((cols, rows))
    open(path, "w").write("\n".join(f"{r} {c}" for r, c in zip(rows[order], cols[order])))
    return path


def test_irregular_code_totals_form(tmp_path):
    """The second register-resident kernel (totals form, kernels_reg2_impl.hpp) on a code that is NOT regular: check nodes
    of degree 5 and 6 (the generic instantiation with its switch over the degree, partly filled blocks), variable nodes
    of degree 2 and 3; 20800 messages = 166 KB, beyond LDS.  Bit for bit against the det oracle, sum-product with and
    without early termination (ratio form, second pass, LLR domain with the saturated form) and min-sum."""
    import libldpc_amd
    rng = np.random.default_rng(11)
    path = make_code_by_degrees(str(tmp_path / "irr.txt"), [2] * 3008 + [3] * 4928, [5] * 560 + [6] * 3000, rng)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    assert (d.nc, d.mc, d.nnz) == (7936, 3560, 20800) == (code.nc, code.mc, code.nnz)
    assert d.residency == "registers" and d.register_form == "totals"
    for x, ms, early, iters in ((3.0, False, True, 30), (2.2, False, True, 12), (3.0, False, False, 40), (2.6, True, True, 20)):
        d.stream_begin("AWGN", 5, x)
        r = d.stream_decode(6, early_term=early, iterations=iters, decoding="BP_MS" if ms else "BP", want=OUT)
        o = code.run_frames("AWGN", x, seed=5, skip=0, count=6, min_sum=ms, early_term=early, iters=iters, math=orc.MATH_DET)
        for k in OUT:
            assert np.array_equal(r[k], o[k].astype(r[k].dtype)), (x, ms, early, k)
    assert 0 < r["iters"].min()


@pytest.mark.parametrize("name,vn,cn,residency", [
    ("lds_regular_3_6", [3] * 600, [6] * 300, "lds"),
    ("lds_mixed", [3] * 500 + [4] * 100 + [6] * 50, [4] * 300 + [5] * 120 + [8] * 50, "lds"),
    ("mem_wide_cn20", [3] * 800, [20] * 120, "memory"),
    ("mem_wide_cn40_24", [3] * 800, [40] * 30 + [24] * 50, "memory"),
])
def test_saturated_check_nodes_in_a_whole_decode(name, vn, cn, residency, tmp_path):
    """Sum-product WITHOUT early termination on codes without degree-1 variable nodes: after convergence the LLRs
    double with every iteration and the check nodes take the saturated form (detmath.h dm_sat_*) — in the register
    tiles of the LDS-resident decoder (through its hand-over) and in the wide-node scratch form of the memory-resident
    one.  Bit for bit against the det oracle, and the run must really have saturated."""
    import libldpc_amd
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    path = make_code_by_degrees(str(tmp_path / f"{name}.txt"), vn, cn, rng)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    assert d.residency == residency, d.residency
    for x, iters in ((10.0, 45), (5.0, 30)):
        d.stream_begin("AWGN", 9, x)
        r = d.stream_decode(6, early_term=False, iterations=iters, decoding="BP", want=OUT)
        o = code.run_frames("AWGN", x, seed=9, skip=0, count=6, early_term=False, iters=iters, math=orc.MATH_DET)
        for k in OUT:
            assert np.array_equal(r[k], o[k].astype(r[k].dtype)), (name, x, k)
    d.stream_begin("AWGN", 9, 10.0)
    r = d.stream_decode(6, early_term=False, iterations=45, decoding="BP", want=("llr_out", "bit_errors"))
    assert (r["bit_errors"] == 0).all() and np.abs(r["llr_out"]).min() > 1e6


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_random_code_bit_exact(case, tmp_path):
    import libldpc_amd
    name, nc, mc, pool, punct, short, skip, residency = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    degs = rng.choice(pool, size=mc)
    path = make_code(str(tmp_path / f"{name}.txt"), nc, mc, degs, rng, punct, short, skip)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    assert (d.nc, d.mc, d.nnz) == (code.nc, code.mc, code.nnz)
    if residency:
        assert d.residency == residency, d.residency
    # channel points chosen so that some frames converge and some do not
    for ch, x, ms, early, iters in (("AWGN", 3.0, False, True, 20), ("AWGN", 1.0, True, True, 15),
                                    ("AWGN", 2.0, False, False, 4), ("AWGN", 9.0, False, True, 20),
                                    ("AWGN", 13.0, False, True, 20), ("BSC", 0.03, False, True, 20),
                                    ("BEC", 0.25, False, True, 20)):
        if name == "bec_beyond_lds" and (ch, x) not in (("BEC", 0.25), ("AWGN", 3.0)):
            continue  # (a 70 000-column code: the oracle side takes seconds per frame)
        d.set_bec_compat(False)
        d.stream_begin(ch, 3, x)
        d.stream_skip(1)
        r = d.stream_decode(5, early_term=early, iterations=iters, decoding="BP_MS" if ms else "BP", want=OUT)
        o = code.run_frames(ch, x, seed=3, skip=1, count=5, min_sum=ms, early_term=early, iters=iters,
                            math=orc.MATH_DET, bec_compat=False)
        for k in OUT:
            assert np.array_equal(r[k], o[k].astype(r[k].dtype)), (name, ch, x, k)


def test_edge_case_batches():
    import libldpc_amd
    d = libldpc_amd.HipDecoder(orc.H_TXT)
    code = orc.Code(orc.H_TXT)
    # zero iterations: the decoder returns 0 and leaves its zero-initialised estimate (decoder.cpp:21-22)
    d.stream_begin("AWGN", 0, -4.0)
    r = d.stream_decode(3, iterations=0, want=OUT)
    o = code.run_frames("AWGN", -4.0, seed=0, count=3, iters=0, math=orc.MATH_DET)
    for k in ("iters", "bit_errors", "hard", "llr_in"):
        assert np.array_equal(r[k], o[k].astype(r[k].dtype)), k
    # one iteration, one frame
    d.stream_begin("AWGN", 0, -4.0)
    r = d.stream_decode(1, iterations=1, want=OUT)
    o = code.run_frames("AWGN", -4.0, seed=0, count=1, iters=1, math=orc.MATH_DET)
    for k in OUT:
        assert np.array_equal(r[k], o[k].astype(r[k].dtype)), k
    # empty batch
    r = d.stream_decode(0, want=("iters",))
    assert r["iters"].shape == (0,)
    assert d.decode_batch(np.zeros((0, d.nc)))["iters"].shape == (0,)


def make_fused_code(path, rng, cn_spec, vn_degs, puncture=(), shorten=()):
    """A random code the fused rule takes (libldpc_amd/csrc/fused_rule.h): check nodes of degree 2..4 with at most one leaf
    each.  cn_spec = [(degree, has_leaf, how many)]; vn_degs = degrees (>= 2) of the other variable nodes, whose sockets
    must equal the check nodes' non-leaf sockets.  Columns: the other variable nodes first, then the leaves; rows shuffled."""
    rows = []
    for deg, leaf, count in cn_spec:
        rows += [(deg, leaf)] * count
    rng.shuffle(rows)
    sockets = sum(d - l for d, l in rows)
    assert sockets == sum(vn_degs), (sockets, sum(vn_degs))
    cols = np.repeat(np.arange(len(vn_degs)), vn_degs)
    owner = np.repeat(np.arange(len(rows)), [d - l for d, l in rows])
    rng.shuffle(cols)
    for _ in range(500):
        key = owner.astype(np.int64) * len(vn_degs) + cols
        _, first = np.unique(key, return_index=True)
        dup = np.setdiff1d(np.arange(key.size), first)
        if dup.size == 0:
            break
        other = rng.integers(0, key.size, dup.size)
        cols[dup], cols[other] = cols[other].copy(), cols[dup].copy()
    else:
        raise AssertionError("could not repair duplicate edges")
    lines, leaf_col = [], len(vn_degs)
    if puncture:
        lines.append(f"puncture [{len(puncture)}]: " + " ".join(map(str, puncture)))
    if shorten:
        lines.append(f"shorten [{len(shorten)}]: " + " ".join(map(str, shorten)))
    for i, (deg, leaf) in enumerate(rows):
        mine = list(cols[owner == i])
        if leaf:
            mine.append(leaf_col)
            leaf_col += 1
        rng.shuffle(mine)  # the leaf and the degree-2 neighbours anywhere in the row's file order
        for c in mine:
            lines.append(f"{i} {c}")
    open(path, "w").write("\n".join(lines))
    return path


FUSED_CASES = [
    # small instantiation shapes besides h.txt: partly filled blocks, a wide block of degree 7, degree-2 check nodes
    ("fused_small", [(2, 0, 30), (3, 0, 50), (3, 1, 40), (4, 0, 20), (4, 1, 70)], [2] * 150 + [7] * 40, (3, 200), ()),
    # the small instantiation on another shape: one leaf class, blocks of degree 3 and 5 through register-held offsets
    ("fused_small_one_class", [(4, 1, 128), (3, 0, 100)], [3] * 128 + [5] * 60, (), ()),
    # the general instantiation: many degree-2 blocks (eight slots per wave), every leaf class (two leaf calls per wave), a
    # block of degree 20 (slot table: wider than the register-held offsets) and blocks of degree 3 and 5 beside a wide one
    ("fused_general", [(2, 0, 60), (3, 0, 300), (3, 1, 120), (4, 0, 200), (4, 1, 160)],
     [2] * 600 + [3] * 200 + [5] * 60 + [20] * 10 + [12] * 10 + [6] * 20, (0, 1, 900), (5,)),
]


@pytest.mark.parametrize("case", FUSED_CASES, ids=[c[0] for c in FUSED_CASES])
def test_fused_form_on_random_codes(case, tmp_path):
    """The fused kernels (kernels_fused.hip) beyond the one code the headline runs: every instantiation (small / general, with
    and without the LLR output), the slot-table path, degree-2 check nodes, partly filled blocks, leaves and degree-2
    neighbours anywhere in a row's file order, punctured and shortened columns — sum-product with early termination (three
    launches: frames escape at high SNR), without (hand-over), min-sum without early termination, BSC; bit for bit against
    the det-mode oracle, which must have taken the fused form too (fused_rule.h is shared; the plan reports it)."""
    import libldpc_amd
    name, cn_spec, vn_degs, punct, short = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    path = make_fused_code(str(tmp_path / f"{name}.txt"), rng, cn_spec, vn_degs, punct, short)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    assert (d.nc, d.mc, d.nnz) == (code.nc, code.mc, code.nnz) and d.residency == "lds"
    assert d.fused_plan()["ok"] == 1, d.fused_plan()
    if name == "fused_general":
        assert d.fused_plan()["vnb"] > 4 and d.fused_plan()["cnl"] == 2, d.fused_plan()
    for ch, x, ms, early, iters in (("AWGN", 2.0, False, True, 25), ("AWGN", 6.0, False, True, 25), ("AWGN", 14.0, False, True, 20),
                                    ("AWGN", 3.0, False, False, 30), ("AWGN", 8.0, False, False, 45), ("AWGN", 2.0, True, False, 20),
                                    ("AWGN", 1.0, True, True, 15), ("BSC", 0.04, False, True, 25)):
        o = code.run_frames(ch, x, seed=4, skip=2, count=6, min_sum=ms, early_term=early, iters=iters, math=orc.MATH_DET)
        for want in (OUT, ("iters", "bit_errors", "hard")):
            d.stream_begin(ch, 4, x)
            d.stream_skip(2)
            r = d.stream_decode(6, early_term=early, iterations=iters, decoding="BP_MS" if ms else "BP", want=want)
            for k in want:
                assert np.array_equal(r[k], o[k].astype(r[k].dtype)), (name, ch, x, ms, early, k, len(want))


@pytest.mark.parametrize("name,vn,cn,residency,form", [
    ("registers_totals", [3] * 8192, [6] * 4096, "registers", "totals"),
    ("registers_messages", [1] * 600 + [3] * 7200, [6] * 3700, "registers", "messages"),
    ("memory", [3] * 16384, [6] * 8192, "memory", None),
])
def test_degree6_shared_reciprocals_three_launches(name, vn, cn, residency, form, tmp_path):
    """Codes the LDS-resident decoder does not take: with early termination their degree-6 check nodes share reciprocals in
    the first launch (detmath.h, dm_cn6_shared), frames in which a product of denominators leaves its range are decoded again
    with separately divided outputs (second launch) and only what leaves the box there goes on to the LLR domain (third).
    Inputs crafted so that all three happen (orc.craft_degree6_overflow), in each of the three decoders for such codes;
    every output bit for bit against the det oracle, which counts the stages."""
    import libldpc_amd
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    path = make_code_by_degrees(str(tmp_path / f"{name}.txt"), vn, cn, rng)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    assert d.residency == residency and (form is None or d.register_form == form), (d.residency, d.register_form)
    kinds = ["odd", "even", "plain", "odd", "odd", "even", "plain", "odd"]
    frames = np.stack([orc.craft_degree6_overflow(code, rng, odd=(k == "odd")) if k != "plain"
                       else np.where(rng.random(code.nc) < 0.02, -1.5, 2.5) for k in kinds])
    r = d.decode_batch(frames, want=("iters", "hard", "llr_out"))
    orc.ratio_stats(reset=True)
    for f, k in enumerate(kinds):
        it, out, hard = code.decode(frames[f], math=orc.MATH_DET)
        assert it == r["iters"][f], (name, f, k)
        assert np.array_equal(hard, r["hard"][f]) and np.array_equal(out, r["llr_out"][f]), (name, f, k)
    done, escaped = orc.ratio_stats()
    assert orc.ratio_second() == kinds.count("odd") + kinds.count("even")  # the first launch gave all crafted frames up
    # far above the threshold: frames converge in the first pass, and the check-node pass the kernels run beyond it (the
    # memory-resident one: always; its outputs are never used) overflows the shared reciprocals — ignored by the rule, and the
    # garbage it writes must not disturb the decisions that ride in the message slots' sign bits
    d.stream_begin("AWGN", 5, 13.5)
    r = d.stream_decode(6, want=OUT)
    o = code.run_frames("AWGN", 13.5, seed=5, count=6, math=orc.MATH_DET)
    for k in OUT:
        assert np.array_equal(r[k], o[k].astype(r[k].dtype)), (name, "13.5 dB", k)
    # the third launch took the even ones (in the code with degree-1 variable nodes some odd ones too), the second finished the rest
    assert escaped >= kinds.count("even") and done + escaped == len(kinds) and done > kinds.count("plain")
    assert escaped == kinds.count("even") or name == "registers_messages"



@pytest.mark.parametrize("compat", [False, True], ids=["defined", "reference_compatible"])
def test_bit_sliced_erasure_decoder_shapes(compat, tmp_path):
    """The bit-sliced erasure decoder (kernels_bec.hip: 32 frames per workgroup) where its special paths meet: variable nodes
    of degree 9, 11 and 20 in blocks that are not multiples of sixteen nodes (four lanes per node, partly filled units),
    degree-1 and degree-2 nodes (the leaf rule in both modes, the swap), check nodes of degree 3..8 (inputs in registers) and
    12 (the two-sweep loop), batches that end inside a group of 32 frames (1, 31, 33, 100) and frames that finish at very
    different passes (the activity mask), with and without early termination: every output `==` the oracle."""
    import libldpc_amd
    rng = np.random.default_rng(5 + compat)
    vn = [1] * 70 + [2] * 301 + [3] * 200 + [9] * 37 + [11] * 20 + [20] * 9
    edges = sum(vn)
    cn = [12] * 20 + [8] * 30 + [5] * 40 + [4] * 100 + [3] * ((edges - 12 * 20 - 8 * 30 - 5 * 40 - 4 * 100) // 3)
    cn[-1] += edges - sum(cn)
    path = make_code_by_degrees(str(tmp_path / "becshapes.txt"), vn, cn, rng)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    d.set_bec_compat(compat)
    for x, early, iters, n, skip in ((0.35, True, 50, 100, 0), (0.45, True, 30, 33, 3), (0.25, False, 7, 31, 1), (0.5, True, 50, 1, 40)):
        d.stream_begin("BEC", 9, x)
        if skip:
            d.stream_skip(skip)
        r = d.stream_decode(n, early_term=early, iterations=iters, want=OUT)
        o = code.run_frames("BEC", x, seed=9, skip=skip, count=n, early_term=early, iters=iters, bec_compat=compat)
        for k in OUT:
            assert np.array_equal(r[k], o[k].astype(r[k].dtype)), (x, n, k)
        if n == 100:
            assert r["iters"].max() - r["iters"].min() >= 3  # the frames of a group really finish apart
