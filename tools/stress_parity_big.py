"""Randomised parity stress for the register-resident decoders (codes beyond LDS: variable nodes of degree 2 and 3, check nodes of degree 5 or 6): early termination near and far above the threshold (all three forms), fixed iterations — against the det-mode oracle.  usage: python tools/stress_parity_big.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, orc, libldpc_amd
from test_gpu_random_codes import make_code_by_degrees
OUT = ("iters", "bit_errors", "hard", "llr_in", "llr_out")
rng = np.random.default_rng(11)
failures = False
for t in range(8):
    n3 = int(rng.integers(5200, 8100)); n2 = int(rng.integers(0, 8190 - n3)) if t % 2 else 0
    vn = [2] * n2 + [3] * n3
    edges = sum(vn); cd = 6 if t % 3 else 5
    cn = [cd] * (edges // cd); rest = edges - sum(cn)
    if rest >= 2: cn.append(rest)
    elif rest == 1: cn[-1] += 1
    path = make_code_by_degrees(f"/tmp/big_{t}.txt", vn, cn, rng)
    code = orc.Code(path); d = libldpc_amd.HipDecoder(path)
    for ch, ms, x, early, iters in (("AWGN", False, float(rng.choice([1.5, 2.5, 4.0])), True, 50), ("AWGN", False, 13.5, True, 50),
                                    ("AWGN", False, 3.0, False, 45), ("AWGN", True, 2.0, bool(t % 2), 20), ("BSC", False, 0.04, True, 30),
                                    ("BEC", False, 0.3, True, 50)):
        seed = int(rng.integers(0, 100))
        d.set_bec_compat(bool(t % 2))
        d.stream_begin(ch, seed, x)
        r = d.stream_decode(4, early_term=early, iterations=iters, decoding="BP_MS" if ms else "BP", want=OUT)
        orc.ratio_stats(reset=True)
        o = code.run_frames(ch, x, seed=seed, count=4, min_sum=ms, early_term=early, iters=iters, math=orc.MATH_DET, bec_compat=bool(t % 2))
        st = orc.ratio_stats()
        bad = [k for k in OUT if not np.array_equal(r[k], o[k].astype(r[k].dtype))]
        if bad:
            print("MISMATCH", t, path, ch, ms, x, early, bad, "stages", st, "second", orc.ratio_second())
            for k in bad:
                diff = np.argwhere(r[k] != o[k].astype(r[k].dtype))
                print("  ", k, "differs at", diff[:6].tolist(), "device", r[k][tuple(diff[0])] if diff.size else None, "oracle", o[k][tuple(diff[0])] if diff.size else None)
            print("   iters device", r["iters"].tolist(), "oracle", o["iters"].tolist(), "bit_errors device", r["bit_errors"].tolist(), "oracle", o["bit_errors"].tolist())
            failures = True
            continue
        print(f"big {t}: nc={code.nc} nnz={code.nnz} {d.residency}/{d.register_form} cn={cd} {ch} ms={ms} x={x} early={early}: stages {st} second {orc.ratio_second()} iters {r['iters'].tolist()} ok", flush=True)
print("big trials passed" if not failures else "FAILURES")
sys.exit(1 if failures else 0)
