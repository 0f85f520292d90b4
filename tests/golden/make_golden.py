#!/usr/bin/env python3
"""Generate the committed golden fixtures from the UNMODIFIED reference.

Runs only where /root/reference exists (this container).  It drives the reference through
the binaries oracle/Makefile builds into oracle/_ref/ (ref_dump = our dumper linked against
the reference's own sources; ldpcsim_ref = the reference CLI; libldpc_ref.so = the reference
C-ABI library) and stores inputs + outputs as data:

  tests/golden/ref_frames.npz   per-frame vectors (llr_in, llr_out, hard, codeword, iters,
                                bit_errors) for a handful of frames per case
  tests/golden/ref_counters.npz per-frame iters / bit_errors for long frame runs
  tests/golden/ref_sim.json     result-file lines of the reference CLI and C-ABI outputs

Usage:  make -C oracle ref && python tests/golden/make_golden.py
"""
import ctypes as ct
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

REF = "/root/reference"
H = os.path.join(HERE, "h.txt")   # byte-identical copies of the reference's test data files
G = os.path.join(HERE, "g.txt")

# name: (G?, channel, decoder, iters, early, seed, x, skip, count)
FRAME_CASES = {
    "awgn_bp_m4": (False, "AWGN", "BP", 50, 1, 0, -4.0, 0, 8),
    "awgn_bp_m4_skip1216": (False, "AWGN", "BP", 50, 1, 0, -4.0, 1216, 2),
    "awgn_ms_m5": (False, "AWGN", "BP_MS", 50, 1, 0, -5.0, 0, 8),
    "awgn_bp_m6_noearly": (False, "AWGN", "BP", 50, 0, 0, -6.0, 0, 4),
    "awgn_ms_m4_noearly_i7": (False, "AWGN", "BP_MS", 7, 0, 3, -4.0, 2, 4),
    "awgn_bp_m45_G_seed7": (True, "AWGN", "BP", 50, 1, 7, -4.5, 0, 6),
    "bsc_bp_024": (False, "BSC", "BP", 50, 1, 0, 0.24, 0, 6),
    "bsc_ms_028_G": (True, "BSC", "BP_MS", 50, 1, 0, 0.28, 0, 4),
    "bec_07": (False, "BEC", "BP", 50, 1, 0, 0.7, 0, 6),
    "bec_08_G": (True, "BEC", "BP", 50, 1, 0, 0.8, 0, 6),
    "bec_09_G_noearly": (True, "BEC", "BP", 50, 0, 5, 0.9, 0, 4),
}

COUNTER_CASES = {
    "awgn_bp_m4": (False, "AWGN", "BP", 50, 1, 0, -4.0, 0, 2000),
    "awgn_ms_m45": (False, "AWGN", "BP_MS", 50, 1, 0, -4.5, 0, 1000),
    "awgn_bp_m5_noearly": (False, "AWGN", "BP", 50, 0, 0, -5.0, 0, 100),
    "bsc_bp_024": (False, "BSC", "BP", 50, 1, 0, 0.24, 0, 300),
    "bec_08_G": (True, "BEC", "BP", 50, 1, 0, 0.8, 0, 500),
    "awgn_bp_m4_G_seed11": (True, "AWGN", "BP", 50, 1, 11, -4.0, 0, 300),
}

# reference CLI runs: args after "codefile output-file"
SIM_CASES = {
    "awgn_bp": ["-4", "-3.99", "1", "-s", "0", "-t", "1", "--max-frames", "2000"],
    "awgn_ms_sweep": ["-6", "-4.4", "0.5", "-s", "1", "-t", "1", "--decoding", "BP_MS", "--max-frames", "400",
                      "--frame-error-count", "30"],
    "bsc": ["0.12", "0.3", "0.04", "--channel", "BSC", "--max-frames", "300", "--frame-error-count", "20"],
    "bec_G": ["0.7", "0.95", "0.1", "--channel", "BEC", "-G", G, "--max-frames", "400", "--frame-error-count", "25"],
    "awgn_bp_noearly_i10": ["-5", "-4.5", "1", "-i", "10", "--no-early-term", "--max-frames", "200",
                            "--frame-error-count", "10"],
}


class decoder_param(ct.Structure):
    _fields_ = [("earlyTerm", ct.c_bool), ("iterations", ct.c_uint32), ("type", ct.c_char_p)]


def cabi_cases(tmp):
    """Outputs of the reference libldpc.so entry points (shared.cpp:9-78)."""
    lib = ct.CDLL(orc.REF_LIB)
    n, m, nct, mct = ct.c_int(), ct.c_int(), ct.c_int(), ct.c_int()
    lib.ldpc_setup(H.encode(), G.encode(), ct.byref(n), ct.byref(m), ct.byref(nct), ct.byref(mct))
    out = {"setup": [n.value, m.value, nct.value, mct.value], "rank": lib.calculate_rank()}
    rng = np.random.default_rng(2024)
    info = rng.integers(0, 2, size=(4, nct.value - mct.value), dtype=np.uint8)
    cws = np.zeros((4, nct.value), np.uint8)
    for i in range(4):
        lib.encode(info[i].ctypes.data_as(ct.c_void_p), cws[i].ctypes.data_as(ct.c_void_p))
    words = rng.integers(0, 2, size=(3, n.value), dtype=np.uint8)
    synd = np.zeros((3, n.value), np.uint8)  # pyLDPC passes an nc-sized buffer
    for i in range(3):
        lib.syndrome(words[i].ctypes.data_as(ct.c_void_p), synd[i].ctypes.data_as(ct.c_void_p))
    # decode(): transmitted-length LLR vectors; BP first (set_param is sticky)
    lib.decode.restype = ct.c_int
    sigma2 = 10 ** 0.4
    llr = 2 * (1 + np.sqrt(sigma2) * rng.standard_normal((4, nct.value))) / sigma2
    dec = {}
    for name, (typ, early, iters, rows) in {"bp_early": (b"BP", True, 50, [0, 1]),
                                            "bp_noearly_i5": (b"BP", False, 5, [2]),
                                            "ms_early": (b"BP_MS", True, 50, [0, 3])}.items():
        o = np.zeros((len(rows), nct.value))
        its = []
        for k, r in enumerate(rows):
            its.append(lib.decode(decoder_param(early, iters, typ), llr[r].ctypes.data_as(ct.c_void_p),
                                  o[k].ctypes.data_as(ct.c_void_p)))
        dec[name] = {"rows": rows, "iters": its, "llr_out": o}
    return out, {"info": info, "cw": cws, "words": words, "synd": synd[:, :m.value], "llr": llr,
                 **{f"dec_{k}_out": v["llr_out"] for k, v in dec.items()}}, \
        {k: {"rows": v["rows"], "iters": v["iters"]} for k, v in dec.items()}


def main():
    assert orc.have_ref(), "build oracle/_ref first: make -C oracle ref"
    assert open(H, "rb").read() == open(os.path.join(REF, "tests/code/h.txt"), "rb").read()
    tmp = tempfile.mkdtemp()
    frames, counters = {}, {}
    for name, (g, ch, dec, it, early, seed, x, skip, cnt) in FRAME_CASES.items():
        r = orc.ref_dump(H, G if g else "", ch, dec, it, early, seed, x, skip, cnt, os.path.join(tmp, "f.bin"))
        for k, v in r.items():
            frames[f"{name}/{k}"] = v
        print(name, r["iters"], r["bit_errors"])
    for name, (g, ch, dec, it, early, seed, x, skip, cnt) in COUNTER_CASES.items():
        r = orc.ref_dump(H, G if g else "", ch, dec, it, early, seed, x, skip, cnt, os.path.join(tmp, "f.bin"))
        counters[f"{name}/iters"] = r["iters"].astype(np.uint8)
        counters[f"{name}/bit_errors"] = r["bit_errors"].astype(np.uint16)
        print(name, int(r["iters"].sum()), int((r["bit_errors"] > 0).sum()))
    sim = {"cases": {k: list(v) for k, v in FRAME_CASES.items()},
           "counter_cases": {k: list(v) for k, v in COUNTER_CASES.items()}, "cli": {}}
    for name, args in SIM_CASES.items():
        outf = os.path.join(tmp, "res.txt")
        if os.path.exists(outf):
            os.remove(outf)
        subprocess.check_call([orc.REF_SIM, H, outf] + args, stdout=subprocess.DEVNULL)
        lines = open(outf).read().splitlines() if os.path.exists(outf) else []
        # drop the wall-clock column (frame_time)
        sim["cli"][name] = {"args": [a if a != G else "<G>" for a in args],
                            "lines": [" ".join(ln.split()[:5]) for ln in lines]}
        print(name, sim["cli"][name]["lines"])
    meta, arrs, decmeta = cabi_cases(tmp)
    sim["cabi"] = {**meta, "decode": decmeta}
    for k, v in arrs.items():
        frames[f"cabi/{k}"] = v
    sim["cpu_has_fma"] = "fma" in open("/proc/cpuinfo").read()
    np.savez_compressed(os.path.join(HERE, "ref_frames.npz"), **frames)
    np.savez_compressed(os.path.join(HERE, "ref_counters.npz"), **counters)
    json.dump(sim, open(os.path.join(HERE, "ref_sim.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
