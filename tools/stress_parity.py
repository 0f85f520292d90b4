"""Randomised parity stress on the GPU box: random codes (degree distributions drawn per trial) through the erasure decoder
(both degree-1 rules, early termination on/off, batches that end inside a 32-frame group) and through sum-product with early
termination at channel points where frames are handed from form to form — every output against the det-mode oracle.
usage: python tools/stress_parity.py [trials] [seed]     (prints one line per trial; exits non-zero on the first mismatch)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import orc
from test_gpu_random_codes import make_code_by_degrees
import libldpc_amd

OUT = ("iters", "bit_errors", "hard", "llr_in", "llr_out")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
for t in range(trials):
    nv = int(rng.integers(300, 1400))
    vn = list(rng.choice([1, 2, 2, 2, 3, 3, 4, 9, 12, 17], size=nv))
    edges = int(sum(vn))
    cd = int(rng.choice([3, 4, 5, 6, 7, 9, 13]))
    cn = [cd] * (edges // cd)
    rest = edges - sum(cn)
    if rest == 1:
        cn[-1] += 1
    elif rest:
        cn.append(rest) if rest >= 2 else None
    path = make_code_by_degrees(f"/tmp/stress_{t}.txt", vn, cn, rng)
    code = orc.Code(path)
    d = libldpc_amd.HipDecoder(path)
    seed = int(rng.integers(0, 1000))
    for compat in (False, True):
        d.set_bec_compat(compat)
        eps = float(rng.uniform(0.15, 0.6)); n = int(rng.choice([1, 31, 33, 64, 97])); early = bool(rng.integers(0, 2)); iters = int(rng.choice([0, 3, 50]))
        d.stream_begin("BEC", seed, eps)
        r = d.stream_decode(n, early_term=early, iterations=iters, want=OUT)
        o = code.run_frames("BEC", eps, seed=seed, count=n, early_term=early, iters=iters, bec_compat=compat)
        for k in OUT:
            if not np.array_equal(r[k], o[k].astype(r[k].dtype)):
                print("MISMATCH BEC", t, path, compat, eps, n, early, iters, k); sys.exit(1)
    # the erasure channel's stream over several calls of odd sizes (groups of 32 frames start afresh in every call)
    d.set_bec_compat(False)
    d.stream_begin("BEC", seed, 0.3)
    sizes = [int(v) for v in rng.choice([1, 7, 32, 33, 65], size=4)]
    o = code.run_frames("BEC", 0.3, seed=seed, count=sum(sizes), bec_compat=False)
    at = 0
    for n in sizes:
        r = d.stream_decode(n, want=OUT)
        for k in OUT:
            if not np.array_equal(r[k], o[k][at:at + n].astype(r[k].dtype)):
                print("MISMATCH BEC stream", t, path, sizes, at, k); sys.exit(1)
        at += n
    # other modes on the same code: min-sum (early termination on / off), sum-product with fixed iterations (hand-over), BSC
    for ch, x, ms, early, iters in (("AWGN", float(rng.choice([1.0, 3.0, 8.0])), True, bool(rng.integers(0, 2)), int(rng.choice([5, 30]))),
                                    ("AWGN", float(rng.choice([2.0, 5.0, 10.0])), False, False, int(rng.choice([6, 40]))),
                                    ("BSC", float(rng.choice([0.02, 0.08])), bool(rng.integers(0, 2)), True, 25)):
        n = int(rng.choice([3, 17]))
        d.stream_begin(ch, seed, x)
        r = d.stream_decode(n, early_term=early, iterations=iters, decoding="BP_MS" if ms else "BP", want=OUT)
        o = code.run_frames(ch, x, seed=seed, count=n, min_sum=ms, early_term=early, iters=iters, math=orc.MATH_DET)
        for k in OUT:
            if not np.array_equal(r[k], o[k].astype(r[k].dtype)):
                print("MISMATCH", ch, t, path, x, ms, early, iters, n, k); sys.exit(1)
    x = float(rng.choice([2.0, 6.0, 9.0, 12.0])); n = int(rng.choice([5, 40]))
    d.stream_begin("AWGN", seed, x)
    r = d.stream_decode(n, want=OUT)
    orc.ratio_stats(reset=True)
    o = code.run_frames("AWGN", x, seed=seed, count=n, math=orc.MATH_DET)
    done, esc = orc.ratio_stats()
    for k in OUT:
        if not np.array_equal(r[k], o[k].astype(r[k].dtype)):
            print("MISMATCH AWGN", t, path, x, n, k); sys.exit(1)
    print(f"trial {t}: nc={code.nc} mc={code.mc} nnz={code.nnz} residency={d.residency} cn_degree={cd} AWGN {x} dB: {done} ratio / {orc.ratio_second()} second / {esc} LLR-domain frames: ok", flush=True)
print("all trials passed")
