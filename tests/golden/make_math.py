#!/usr/bin/env python3
"""Fixture for the arithmetic tests (tests/test_gpu_math.py): operands and the values the functions of
libldpc_amd/csrc/detmath.h / device_cn.hpp are MEANT to return, computed without that header — glibc libm in binary64
for exp, log and the reference's jacobian expression (src/decoding/decoder.h:12-15), x87 long double for the rational
expressions and for the plain divided forward/backward check-node recursion (src/decoding/decoder.cpp:31-44).

The values come from oracle/liboracle.so's orc_math_ref (oracle/ldpc_oracle.c) on operands drawn by
tests/orc.py:math_points(fn, n, seed=2026): 4096 points per scalar function, 1024 rows per check-node function, box
edges (|L| = 166, 600, 700, 709) and special operands included.  usage: python tests/golden/make_math.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

SEED = 2026


def main():
    out = {}
    for fn in orc.MATH_FNS:
        a, b = orc.math_points(fn, 1024 if fn.startswith("cn_") else 4096, SEED)
        with np.errstate(all="ignore"):
            ref = orc.math_eval(fn, a, b)
        out[f"{fn}/a"] = a
        if b is not None:
            out[f"{fn}/b"] = b
        out[f"{fn}/ref"] = ref
    np.savez_compressed(os.path.join(HERE, "math_ref.npz"), **out)
    print("functions", len(orc.MATH_FNS), "bytes", os.path.getsize(os.path.join(HERE, "math_ref.npz")))


if __name__ == "__main__":
    main()
