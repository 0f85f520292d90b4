#!/usr/bin/env python3
"""Fixture: the REFERENCE's own Python wrapper (pyLDPC/ldpc.py, imported from the reference checkout, unmodified)
driving OUR library over its ctypes boundary — `LDPC(pc, gen, lib=<libldpc_amd/libldpc.so>)` — for the entry points
that need no GPU (ldpc_setup, calculate_rank, encode, syndrome; src/shared.cpp:11-24, 32-45, 67-77), next to the same
calls against the reference's library (oracle/_ref/libldpc_ref.so).  The outputs of both are recorded in
tests/golden/pyldpc_host.json; tests/test_host.py checks them equal and checks libldpc_amd.LDPC (our mirror of the
wrapper) against them.

Runs in the build container only (it imports from /root/reference, which does not travel); the fixture travels.
Each library is loaded in its own process: both export the same global symbols.
usage: python tests/golden/make_pyldpc.py
"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LDPC_REFERENCE", "/root/reference")
H, G = os.path.join(HERE, "h.txt"), os.path.join(HERE, "g.txt")


def drive(lib, q):
    sys.path.insert(0, REF)
    from pyLDPC.ldpc import LDPC  # the reference's wrapper, as it stands
    c = LDPC(H, G, lib=lib)
    rng = np.random.default_rng(2026)
    out = {"dims": [c.n, c.m, c.nct, c.mct, c.k, c.kct], "rank": int(c.rank()), "encode": [], "syndrome": []}
    for _ in range(6):
        u = rng.integers(0, 2, c.kct)
        cw = c.encode(u)
        out["encode"].append({"info": u.tolist(), "codeword": [int(v) for v in cw]})
    for _ in range(4):
        w = rng.integers(0, 2, c.n)
        out["syndrome"].append({"word": w.tolist(), "syndrome": [int(v) for v in c.syndrome(w)]})
    q.put(out)


def main():
    ours = os.path.join(ROOT, "libldpc_amd", "libldpc.so")
    ref = os.path.join(ROOT, "oracle", "_ref", "libldpc_ref.so")
    assert os.path.isdir(os.path.join(REF, "pyLDPC")), "reference checkout not found"
    res = {}
    ctx = mp.get_context("spawn")
    for name, lib in (("libldpc_amd", ours), ("reference", ref)):
        q = ctx.Queue()
        p = ctx.Process(target=drive, args=(lib, q))
        p.start()
        res[name] = q.get(timeout=300)
        p.join()
    assert res["libldpc_amd"] == res["reference"], "our library and the reference's disagree behind pyLDPC"
    json.dump({"note": "pyLDPC/ldpc.py of the reference over libldpc_amd/libldpc.so == over the reference's libldpc.so",
               "wrapper": "pyLDPC.ldpc.LDPC (reference, unmodified)", **res["libldpc_amd"]},
              open(os.path.join(HERE, "pyldpc_host.json"), "w"))
    print("ok:", res["libldpc_amd"]["dims"], "rank", res["libldpc_amd"]["rank"])


if __name__ == "__main__":
    main()
