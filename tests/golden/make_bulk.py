#!/usr/bin/env python3
"""Bulk fixture: per-frame iteration count and bit-error count of the REFERENCE for the first 100 000 frames of the
headline workload (h.txt, AWGN -4 dB, BP, 50 iterations, early termination, seed 0, all-zero codeword).

Produced by oracle/_ref/ref_dump (our dumper linked against the unmodified reference sources, `make -C oracle ref`)
in chunks of 1000 frames, eight processes at a time; only the two counters are kept (tests/golden/ref_bulk.npz).
usage: python tests/golden/make_bulk.py
"""
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

N, CHUNK = 100000, 1000
CASE = ("AWGN", "BP", 50, 1, 0, -4.0)  # channel, decoder, iterations, early termination, seed, SNR (dB)


def chunk(k):
    ch, dec, it, early, seed, x = CASE
    tmp = tempfile.mkdtemp()
    r = orc.ref_dump(os.path.join(HERE, "h.txt"), "", ch, dec, it, early, seed, x, k * CHUNK, CHUNK, os.path.join(tmp, "f.bin"))
    os.remove(os.path.join(tmp, "f.bin"))
    return r["iters"].astype(np.uint8), r["bit_errors"].astype(np.uint16)


def main():
    assert orc.have_ref(), "build oracle/_ref first: make -C oracle ref"
    with mp.Pool(8) as pool:
        res = pool.map(chunk, range(N // CHUNK))
    it = np.concatenate([r[0] for r in res])
    be = np.concatenate([r[1] for r in res])
    np.savez_compressed(os.path.join(HERE, "ref_bulk.npz"), iters=it, bit_errors=be,
                        case=np.array([str(CASE)]))
    print("frames", N, "frame errors", int((be > 0).sum()), "not converged", int((it >= 50).sum()), "iters", int(it.sum()))


if __name__ == "__main__":
    main()
