#!/usr/bin/env python3
"""Reference fixtures for decoder shapes the other fixtures do not reach (tests/golden/ref_shapes.npz).

h.txt, its shortened variant and the (3,6) n=8192 code pin the LDS-resident kernel and the regular-code instantiation of
the register-resident one.  Two more kernels are pinned here by the UNMODIFIED reference (oracle/_ref/ref_dump, i.e. the
reference's own src/decoding + src/sim compiled by oracle/Makefile), not only by the arithmetic header they share with
the det-mode oracle:

  wide20   800 x 120, every check node of weight 20  -> memory-resident decoder, wide-node scratch form (cn_wide)
  irr      7936 x 3560, check nodes of weight 5 and 6, variable nodes of degree 2 and 3 -> register-resident decoder,
           totals form, the generic (irregular) instantiation

The code files are not committed: tests/test_gpu_random_codes.make_code_by_degrees regenerates them from a seeded
generator and the fixture records their sha256.  Stored per case: iters, bit_errors, hard decisions (bit-packed), llr_out.

Usage (where /root/reference exists):  make -C oracle ref && python tests/golden/make_shapes.py
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402
from test_gpu_random_codes import make_code_by_degrees  # noqa: E402

# name: (vn degrees, cn degrees, generator seed)
CODES = {
    "wide20": ([3] * 800, [20] * 120, 20),
    "irr": ([2] * 3008 + [3] * 4928, [5] * 560 + [6] * 3000, 11),
}
# name: (code, channel, decoder, iters, early, seed, x, skip, count)
CASES = {
    "wide20/awgn_bp_5p5": ("wide20", "AWGN", "BP", 50, 1, 1, 5.5, 0, 12),
    "wide20/awgn_bp_6p0_noearly_i20": ("wide20", "AWGN", "BP", 20, 0, 2, 6.0, 1, 4),
    "wide20/awgn_ms_5p7": ("wide20", "AWGN", "BP_MS", 50, 1, 1, 5.7, 0, 10),
    "wide20/bsc_bp_0012": ("wide20", "BSC", "BP", 50, 1, 0, 0.012, 0, 8),
    "irr/awgn_bp_3p0": ("irr", "AWGN", "BP", 30, 1, 5, 3.0, 0, 3),
    "irr/awgn_bp_2p2_i12": ("irr", "AWGN", "BP", 12, 1, 5, 2.2, 0, 2),
    "irr/awgn_bp_3p0_noearly_i40": ("irr", "AWGN", "BP", 40, 0, 5, 3.0, 0, 2),
    "irr/awgn_ms_2p6": ("irr", "AWGN", "BP_MS", 20, 1, 5, 2.6, 0, 2),
}


def code_file(name, directory):
    vn, cn, seed = CODES[name]
    return make_code_by_degrees(os.path.join(directory, f"shape_{name}.txt"), vn, cn, np.random.default_rng(seed))


def main():
    assert orc.have_ref(), "build oracle/_ref first: make -C oracle ref"
    tmp = tempfile.mkdtemp()
    out = {}
    paths = {}
    for name in CODES:
        paths[name] = code_file(name, tmp)
        out[f"sha256/{name}"] = np.frombuffer(hashlib.sha256(open(paths[name], "rb").read()).digest(), np.uint8)
    for key, (code, ch, dec, it, early, seed, x, skip, cnt) in CASES.items():
        r = orc.ref_dump(paths[code], "", ch, dec, it, early, seed, x, skip, cnt, os.path.join(tmp, "f.bin"))
        out[f"{key}/iters"] = r["iters"].astype(np.int32)
        out[f"{key}/bit_errors"] = r["bit_errors"].astype(np.int32)
        out[f"{key}/hard_packed"] = np.packbits(r["hard"], axis=1)
        out[f"{key}/llr_out"] = r["llr_out"]
        print(key, r["iters"], r["bit_errors"])
    np.savez_compressed(os.path.join(HERE, "ref_shapes.npz"), **out)


if __name__ == "__main__":
    main()
