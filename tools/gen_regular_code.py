#!/usr/bin/env python3
"""Deterministic (dv,dc)-regular LDPC parity-check matrix in the reference's file format.

BASELINE.json configs[3]: n~8k regular LDPC, row weight 6 -> (3,6)-regular, nc=8192, mc=4096, nnz=24576,
seed 1.  Configuration model: the nc*dv variable-node sockets are shuffled with a
self-contained xorshift64* generator (no dependence on numpy/python RNG versions) and dealt dc at a time to
the check nodes; parallel edges are removed by swapping sockets with later positions.  Rows are written in
row-major order, columns ascending inside a row, no header, no trailing blank line (the reference parser
would read a blank line as an edge).

usage: gen_regular_code.py OUT [--nc 8192] [--dv 3] [--dc 6] [--seed 1]
"""
import argparse
import hashlib

MASK = (1 << 64) - 1


class XorShift64Star:
    def __init__(self, seed):
        self.s = (seed * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & MASK or 1

    def next(self):
        x = self.s
        x ^= x >> 12
        x ^= (x << 25) & MASK
        x ^= x >> 27
        self.s = x
        return (x * 0x2545F4914F6CDD1D) & MASK

    def below(self, n):
        # rejection sampling: unbiased
        lim = MASK - (MASK + 1) % n
        while True:
            v = self.next()
            if v <= lim:
                return v % n


def generate(nc=8192, dv=3, dc=6, seed=1):
    assert (nc * dv) % dc == 0
    mc = nc * dv // dc
    rng = XorShift64Star(seed)
    sockets = [v for v in range(nc) for _ in range(dv)]
    for i in range(len(sockets) - 1, 0, -1):  # Fisher-Yates
        j = rng.below(i + 1)
        sockets[i], sockets[j] = sockets[j], sockets[i]
    # repair parallel edges: a duplicate inside a check's dc sockets is swapped with a random socket elsewhere
    for _ in range(100):
        dirty = False
        for c in range(mc):
            row = sockets[c * dc:(c + 1) * dc]
            seen = set()
            for k, v in enumerate(row):
                if v in seen:
                    dirty = True
                    while True:
                        j = rng.below(len(sockets))
                        c2 = j // dc
                        if c2 == c:
                            continue
                        other = sockets[c2 * dc:(c2 + 1) * dc]
                        if sockets[j] in row or v in other:
                            continue
                        sockets[c * dc + k], sockets[j] = sockets[j], sockets[c * dc + k]
                        row = sockets[c * dc:(c + 1) * dc]
                        break
                seen.add(sockets[c * dc + k])
        if not dirty:
            break
    else:
        raise RuntimeError("could not remove parallel edges")
    lines = []
    for c in range(mc):
        for v in sorted(sockets[c * dc:(c + 1) * dc]):
            lines.append(f"{c} {v}")
    return "\n".join(lines)  # no trailing newline-only line


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--nc", type=int, default=8192)
    ap.add_argument("--dv", type=int, default=3)
    ap.add_argument("--dc", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    text = generate(a.nc, a.dv, a.dc, a.seed)
    with open(a.out, "w") as f:
        f.write(text)
    print(a.out, "sha256", hashlib.sha256(text.encode()).hexdigest(), "edges", text.count("\n") + 1)
