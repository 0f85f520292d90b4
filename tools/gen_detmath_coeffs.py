#!/usr/bin/env python3
"""Generate the polynomial coefficients used by libldpc_amd/csrc/detmath.h.

exp:  exp(r) = 1 + r + r^2 * G(r),  |r| <= ln2/2,  G fitted at Chebyshev nodes.
log:  log(1+f) = f - f^2/2 + s*(f^2/2 + R(z)),  s = f/(2+f), z = s^2,
      R(z) = z * Q(z),  Q(z) ~ 2/3 + 2/5 z + 2/7 z^2 + ...  on z in [0, ZMAX].
The fits are interpolation at Chebyshev nodes (near-minimax); the script prints
C hex-float literals and the max relative error measured on a dense grid.
"""
import mpmath as mp

mp.mp.prec = 200


def cheb_fit(fn, a, b, deg):
    n = deg + 1
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        y[i] = fn(x)
    c = mp.lu_solve(A, y)
    return [c[i] for i in range(n)]


def to_double(x):
    return float(mp.nstr(x, 40))


def G(r):
    if abs(r) < mp.mpf(2) ** -60:
        return mp.mpf(1) / 2 + r / 6
    return (mp.exp(r) - 1 - r) / (r * r)


def Q(z):
    if z < mp.mpf(2) ** -80:
        return mp.mpf(2) / 3
    s = mp.sqrt(z)
    # log((1+s)/(1-s)) = 2s + s*R(z)  ->  R = log(..)/s - 2 ; Q = R/z
    return (mp.log((1 + s) / (1 - s)) / s - 2) / z


def report(name, coeffs, fn, a, b, weight=lambda x: 1):
    cd = [to_double(c) for c in coeffs]
    worst = mp.mpf(0)
    N = 4001
    for i in range(N):
        x = a + (b - a) * mp.mpf(i) / (N - 1)
        p = mp.mpf(0)
        for c in reversed(cd):
            p = p * x + mp.mpf(c)
        e = abs(p - fn(x)) * weight(x)
        worst = max(worst, e)
    print(f"/* {name}: degree {len(cd)-1}, max weighted abs err {mp.nstr(worst, 5)} (2^{mp.nstr(mp.log(worst, 2), 5)}) */")
    for i, c in enumerate(cd):
        print(f"#define DM_{name}{i} {c.hex()}")


if __name__ == "__main__":
    half_ln2 = mp.log(2) / 2
    # error of exp = r^2 * err(G); weight by r^2 (result ~ 1)
    for deg in (9, 10):
        report(f"EXP_G{deg}_", cheb_fit(G, -half_ln2, half_ln2, deg), G, -half_ln2, half_ln2, lambda r: r * r)
    smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
    zmax = smax * smax * mp.mpf("1.0001")
    # error of log ~ s*z*err(Q) relative to log ~ 2s  -> weight z/2
    for deg in (6, 7):
        report(f"LOG_Q{deg}_", cheb_fit(Q, 0, zmax, deg), Q, mp.mpf(0), zmax, lambda z: z / 2)
    print("ln2     ", float(mp.log(2)).hex())
    ln2 = mp.log(2)
    hi = float(mp.nstr(ln2, 40))
    print("ln2_lo  ", float(mp.nstr(ln2 - mp.mpf(hi), 40)).hex())
    # split with trailing zeros (32 significant bits) for exact k*hi
    import math
    hi32 = math.ldexp(math.floor(math.ldexp(hi, 32)), -32)
    print("ln2_hi32", hi32.hex(), " lo", float(mp.nstr(ln2 - mp.mpf(hi32), 40)).hex())
    print("1/ln2   ", float(mp.nstr(1 / ln2, 40)).hex())
