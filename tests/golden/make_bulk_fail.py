#!/usr/bin/env python3
"""Fixture: the REFERENCE's hard decisions on the frames it fails among the first 100 000 frames of the headline
workload (the frames with bit_errors > 0 in ref_bulk.npz: 132 of them, 125 not converged + 7 wrong codewords).

Produced by oracle/_ref/ref_dump (our dumper linked against the unmodified reference sources, `make -C oracle ref`),
one call per failing frame (the dumper's skip replays the channel only).  Stored: frame indices and the hard
decisions packed 8 per byte (tests/golden/ref_bulk_fail.npz).
usage: python tests/golden/make_bulk_fail.py
"""
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402
from make_bulk import CASE  # noqa: E402


def one(f):
    ch, dec, it, early, seed, x = CASE
    tmp = tempfile.mkdtemp()
    r = orc.ref_dump(os.path.join(HERE, "h.txt"), "", ch, dec, it, early, seed, x, int(f), 1, os.path.join(tmp, "f.bin"))
    os.remove(os.path.join(tmp, "f.bin"))
    return int(r["iters"][0]), int(r["bit_errors"][0]), np.packbits(r["hard"][0])


def main():
    assert orc.have_ref(), "build oracle/_ref first: make -C oracle ref"
    bulk = np.load(os.path.join(HERE, "ref_bulk.npz"))
    frames = np.flatnonzero(bulk["bit_errors"] > 0).astype(np.uint32)
    with mp.Pool(8) as pool:
        res = pool.map(one, frames)
    for f, (it, be, _) in zip(frames, res):
        assert it == bulk["iters"][f] and be == bulk["bit_errors"][f], f
    np.savez_compressed(os.path.join(HERE, "ref_bulk_fail.npz"), frames=frames, hard_packed=np.stack([r[2] for r in res]),
                        nc=np.array([1152]))
    print("failing frames", len(frames))


if __name__ == "__main__":
    main()
