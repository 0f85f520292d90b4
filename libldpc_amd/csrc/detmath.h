/*
 * detmath.h — deterministic fp64 exp / log for the BP box-plus kernel.
 *
 * Why this exists: the reference's box-plus `jacobian()` (src/decoding/decoder.h:12-15)
 * and its AWGN noise generator (libstdc++ normal_distribution, polar method) call glibc
 * `exp` / `log`.  glibc's results are not reproducible on a GPU (table driven, FMA usage
 * selected by ifunc per CPU).  These two routines are pure IEEE-754 binary64 arithmetic
 * (add, mul, fma, div, integer bit moves) in a fixed evaluation order, so the HIP kernels
 * and the CPU oracle built with ORC_MATH_DET produce bit-identical results, while staying
 * within ~1 ulp of the correctly rounded value (and hence of glibc).
 *
 * Compile every translation unit that includes this header with -ffp-contract=off:
 * all fused operations are written explicitly as fma().
 *
 * Coefficients come from tools/gen_detmath_coeffs.py (Chebyshev-node fits, errors there).
 */
#ifndef LDPC_AMD_DETMATH_H
#define LDPC_AMD_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DM_FN __host__ __device__ static inline
#define DM_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#include <math.h>
#include <string.h>
#define DM_FN static inline
#define DM_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif

DM_FN uint64_t dm_bits(double x)
{
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}

DM_FN double dm_from_bits(uint64_t u)
{
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}

/* exp(r) = 1 + r + r^2 G(r), |r| <= ln2/2, relative error 2^-56.1 */
#define DM_EXP_G0 0x1.0000000000001p-1
#define DM_EXP_G1 0x1.5555555555556p-3
#define DM_EXP_G2 0x1.5555555553d68p-5
#define DM_EXP_G3 0x1.11111111109b5p-7
#define DM_EXP_G4 0x1.6c16c17889ef1p-10
#define DM_EXP_G5 0x1.a01a01a7c2efep-13
#define DM_EXP_G6 0x1.a019b9149a41cp-16
#define DM_EXP_G7 0x1.71de0db2f6b19p-19
#define DM_EXP_G8 0x1.28917c89a43a7p-22
#define DM_EXP_G9 0x1.af389ecfc4b9cp-26

#define DM_INV_LN2 0x1.71547652b82fep+0
#define DM_LN2_HI 0x1.62e42fefa39efp-1
#define DM_LN2_LO 0x1.abc9e3b39803fp-56
#define DM_RND_MAGIC 0x1.8p52

/*
 * dm_exp: x = k ln2 + r, exp(x) = 2^k (1 + r + r^2 G(r)).
 * Finite for x in [-745.2, 709.78]; 0 below, +inf above; NaN propagates.
 */
DM_FN double dm_exp(double x)
{
    if (!(x == x))
        return x;
    if (x > 0x1.62e42fefa39efp+9)
        return __builtin_huge_val();
    if (x < -0x1.74910d52d3052p+9)
        return 0.0;
    double t = x * DM_INV_LN2;
    double kd = (t + DM_RND_MAGIC) - DM_RND_MAGIC; /* round to nearest even */
    double r = DM_FMA(kd, -DM_LN2_HI, x);
    r = DM_FMA(kd, -DM_LN2_LO, r);
    double g = DM_EXP_G9;
    g = DM_FMA(g, r, DM_EXP_G8);
    g = DM_FMA(g, r, DM_EXP_G7);
    g = DM_FMA(g, r, DM_EXP_G6);
    g = DM_FMA(g, r, DM_EXP_G5);
    g = DM_FMA(g, r, DM_EXP_G4);
    g = DM_FMA(g, r, DM_EXP_G3);
    g = DM_FMA(g, r, DM_EXP_G2);
    g = DM_FMA(g, r, DM_EXP_G1);
    g = DM_FMA(g, r, DM_EXP_G0);
    double r2 = r * r;
    double s = DM_FMA(r2, g, r);
    double p = 1.0 + s;
    int64_t k = (int64_t)kd;
    if (k >= -1021 && k <= 1023)
        return p * dm_from_bits((uint64_t)(k + 1023) << 52);
    if (k > 1023) /* k == 1024: p < 1 here, split the scale */
        return (p * 0x1p1023) * 2.0;
    /* gradual underflow: two exact power-of-two scalings, one rounding */
    return (p * dm_from_bits((uint64_t)(k + 1023 + 1000) << 52)) * 0x1p-1000;
}

/* log(1+f) = f - f^2/2 + s (f^2/2 + z Q(z)), s = f/(2+f), z = s^2, rel. error 2^-57.6 */
#define DM_LOG_Q0 0x1.5555555555558p-1
#define DM_LOG_Q1 0x1.99999999952d7p-2
#define DM_LOG_Q2 0x1.2492492df281ap-2
#define DM_LOG_Q3 0x1.c71c62e3f11e6p-3
#define DM_LOG_Q4 0x1.7462b51cb66b1p-3
#define DM_LOG_Q5 0x1.39fe51a7c18f9p-3
#define DM_LOG_Q6 0x1.2b5900de53b32p-3

#define DM_LN2_HI32 0x1.62e42fee00000p-1
#define DM_LN2_LO32 0x1.a39ef35793c76p-33
#define DM_SQRT_HALF_BITS 0x3FE6A09E667F3BCDull

/*
 * dm_log: x = 2^k m, m in [sqrt(1/2), sqrt(2)), log x = k ln2 + log(1+f), f = m-1.
 * x < 0 -> NaN, x == 0 -> -inf, +inf -> +inf, subnormals handled.
 */
DM_FN double dm_log(double x)
{
    uint64_t ix = dm_bits(x);
    int64_t kadj = 0;
    if (ix - 0x0010000000000000ull >= 0x7FE0000000000000ull) /* not positive normal */
    {
        if ((ix << 1) == 0)
            return -__builtin_huge_val();
        if (ix == 0x7FF0000000000000ull)
            return x;
        if ((ix >> 63) || !(x == x))
            return (x - x) / 0.0 * 0.0 + __builtin_nan("");
        x = x * 0x1p54; /* subnormal */
        ix = dm_bits(x);
        kadj = -54;
    }
    uint64_t tmp = ix - DM_SQRT_HALF_BITS;
    int64_t k = (int64_t)tmp >> 52;
    double m = dm_from_bits(ix - ((uint64_t)k << 52));
    k += kadj;
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double q = DM_LOG_Q6;
    q = DM_FMA(q, z, DM_LOG_Q5);
    q = DM_FMA(q, z, DM_LOG_Q4);
    q = DM_FMA(q, z, DM_LOG_Q3);
    q = DM_FMA(q, z, DM_LOG_Q2);
    q = DM_FMA(q, z, DM_LOG_Q1);
    q = DM_FMA(q, z, DM_LOG_Q0);
    double hfsq = 0.5 * f * f;
    double t = DM_FMA(z, q, hfsq); /* hfsq + R */
    double dk = (double)k;
    double lo = DM_FMA(s, t, dk * DM_LN2_LO32);
    return DM_FMA(dk, DM_LN2_HI32, f - (hfsq - lo));
}

/* ------------------------------------------------------------------------------------------------
 * dm_boxplus — the sum-product box-plus of the reference (src/decoding/decoder.h:12-15)
 *
 *     jacobian(x,y) = sign(x) sign(y) min(|x|,|y|) + log( (1 + e^-|x+y|) / (1 + e^-|x-y|) )
 *
 * evaluated in the reference's own order (two exponentials, two additions, one IEEE division, one
 * logarithm, one addition) with an exp/log pair specialised for this call site and written for the
 * GPU's instruction mix: branch-free, no table for exp, a 64-entry {1/c, log c} table for log so that
 * no second division is needed.
 *   dm_boxplus_exp(t) = e^-t for t >= 0, relative error < 1.1 ulp; arguments beyond 700 are clamped
 *                       (only 1 + e^-t is ever formed, and 1 + e^-700 == 1).
 *   dm_boxplus_log(q) = log q for q in [1/2, 2], ABSOLUTE error < 1.5e-16 (q itself carries an absolute
 *                       rounding error of 1.1e-16 from the division, in the reference too).
 * Both are pure binary64 arithmetic with explicit fma: the CPU oracle (ORC_MATH_DET) and the HIP
 * kernels produce identical bits.
 * ------------------------------------------------------------------------------------------------ */
#include "detmath_tables.h"

#if defined(__HIPCC__)
__device__ static const double dm_bpl_table_dev[2 << DM_BPL_NBITS] __attribute__((aligned(16))) = {DM_BPL_TABLE};
#endif
static const double dm_bpl_table_host[2 << DM_BPL_NBITS] __attribute__((aligned(16))) = {DM_BPL_TABLE};

DM_FN double dm_boxplus_exp(double t)
{
    double x = -__builtin_fmin(t, 700.0);
    double z = x * DM_INV_LN2;
#if defined(__HIP_DEVICE_COMPILE__)
    double kd = __builtin_rint(z); /* v_rndne_f64: same value as the add/subtract form below */
#else
    double kd = (z + DM_RND_MAGIC) - DM_RND_MAGIC;
#endif
    double r = DM_FMA(kd, -DM_LN2_HI, x);
    r = DM_FMA(kd, -DM_LN2_LO, r);
    double g = DM_EXP_G9;
    g = DM_FMA(g, r, DM_EXP_G8);
    g = DM_FMA(g, r, DM_EXP_G7);
    g = DM_FMA(g, r, DM_EXP_G6);
    g = DM_FMA(g, r, DM_EXP_G5);
    g = DM_FMA(g, r, DM_EXP_G4);
    g = DM_FMA(g, r, DM_EXP_G3);
    g = DM_FMA(g, r, DM_EXP_G2);
    g = DM_FMA(g, r, DM_EXP_G1);
    g = DM_FMA(g, r, DM_EXP_G0);
    double r2 = r * r;
    double s = DM_FMA(r2, g, r);
    double p = 1.0 + s;
    int k = (int)kd; /* -1010 <= k <= 0: the scaled result is a normal number */
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_ldexp(p, k); /* v_ldexp_f64: exact scaling, same value as the multiply below */
#else
    return p * dm_from_bits((uint64_t)(k + 1023) << 52);
#endif
}

/* e^x for |x| <= 700 (arguments beyond are clamped), branch-free: the same reduction, polynomial and scaling as
   dm_exp — identical bits wherever both are defined — without its special cases and its 64-bit integer detour.
   Used for lambda(L_ch) = e^-L_ch at the head of the likelihood-ratio form (|L_ch| <= 166 there, or the frame has
   already escaped). */
DM_FN double dm_exp_clamped(double x)
{
    x = __builtin_fmin(__builtin_fmax(x, -700.0), 700.0);
    double z = x * DM_INV_LN2;
#if defined(__HIP_DEVICE_COMPILE__)
    double kd = __builtin_rint(z); /* v_rndne_f64: same value as the add/subtract form below */
#else
    double kd = (z + DM_RND_MAGIC) - DM_RND_MAGIC;
#endif
    double r = DM_FMA(kd, -DM_LN2_HI, x);
    r = DM_FMA(kd, -DM_LN2_LO, r);
    double g = DM_EXP_G9;
    g = DM_FMA(g, r, DM_EXP_G8);
    g = DM_FMA(g, r, DM_EXP_G7);
    g = DM_FMA(g, r, DM_EXP_G6);
    g = DM_FMA(g, r, DM_EXP_G5);
    g = DM_FMA(g, r, DM_EXP_G4);
    g = DM_FMA(g, r, DM_EXP_G3);
    g = DM_FMA(g, r, DM_EXP_G2);
    g = DM_FMA(g, r, DM_EXP_G1);
    g = DM_FMA(g, r, DM_EXP_G0);
    double r2 = r * r;
    double s = DM_FMA(r2, g, r);
    double p = 1.0 + s;
    int k = (int)kd; /* |k| <= 1010: the scale factor is a normal number */
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_ldexp(p, k); /* v_ldexp_f64: exact scaling, same value as the multiply below */
#else
    return p * dm_from_bits((uint64_t)(k + 1023) << 52);
#endif
}

DM_FN double dm_boxplus_log(double q)
{
    uint64_t ix = dm_bits(q);
    uint32_t hi = (uint32_t)(ix >> 32);
    uint32_t tmp = hi - DM_BPL_OFF_HI; /* the low word of the offset is zero */
    int32_t k = (int32_t)tmp >> 20;
    uint32_t i = (tmp >> (20 - DM_BPL_NBITS)) & ((1u << DM_BPL_NBITS) - 1u);
    uint32_t zhi = hi - (tmp & 0xFFF00000u);
    double z = dm_from_bits(((uint64_t)zhi << 32) | (uint32_t)ix);
#if defined(__HIP_DEVICE_COMPILE__)
    const double2 tc = *reinterpret_cast<const double2 *>(dm_bpl_table_dev + 2 * i);
    double invc = tc.x, logc = tc.y;
#else
    double invc = dm_bpl_table_host[2 * i], logc = dm_bpl_table_host[2 * i + 1];
#endif
    double r = DM_FMA(z, invc, -1.0);
    double r2 = r * r;
    double p = DM_BPL_P5;
    p = DM_FMA(p, r, DM_BPL_P4);
    p = DM_FMA(p, r, DM_BPL_P3);
    p = DM_FMA(p, r, DM_BPL_P2);
    p = DM_FMA(p, r, DM_BPL_P1);
    p = DM_FMA(p, r, DM_BPL_P0);
    double w = DM_FMA((double)k, DM_LN2_HI, logc);
    double l = DM_FMA(r2, p, r);
    return w + l;
}

DM_FN double dm_ratio_div(double a, double b); /* below: the IEEE quotient for positive in-range operands */

DM_FN double dm_boxplus(double x, double y)
{
    double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    double mn = __builtin_fmin(ax, ay); /* == std::min(|x|, |y|) for every non-NaN pair */
    uint64_t sgn = (dm_bits(x) ^ dm_bits(y)) & 0x8000000000000000ull;
    double m = dm_from_bits(dm_bits(mn) | sgn); /* sign(x) sign(y) min: -0.0 when the signs differ and min is 0 */
    double num = 1.0 + dm_boxplus_exp(__builtin_fabs(x + y));
    double den = 1.0 + dm_boxplus_exp(__builtin_fabs(x - y));
    return m + dm_boxplus_log(dm_ratio_div(num, den)); /* both in [1, 2] */
}

/* ------------------------------------------------------------------------------------------------
 * Check-node form of the box-plus: the forward/backward recursion carried in the variable E = e^-|L|.
 *
 * With E(v) = e^-|v| and s(v) = sign(v), the reference's box-plus (decoder.h:12-15)
 *
 *     x [+] y = s(x) s(y) min(|x|,|y|) + log( (1 + e^-|x+y|) / (1 + e^-|x-y|) )
 *
 * is the same real function as        |x [+] y| = -log( (E(x) + E(y)) / (1 + E(x) E(y)) ),
 *                                   s(x [+] y) = s(x) s(y)
 * (write a = |x|, b = |y|: min(a,b) + log((1+e^-(a+b))/(1+e^-|a-b|)) = log((1+e^-(a+b))/(e^-a+e^-b)) ).
 * So a partial result of the recursion decoder.cpp:31-44 that is only fed into further box-pluses never
 * has to leave the E domain: it is carried as (sign, E) and combined with
 *
 *     dm_e_combine(Ex, Ey) = (Ex + Ey) / (1 + Ex Ey)             one fma, one add, one IEEE division
 *
 * A degree-d check node then costs d exponentials (its inputs), 3(d-2) combines and d logarithms (its
 * outputs) instead of 2 exponentials, a division and a logarithm for each of its 3(d-2) box-pluses.
 * Same recursion order as the reference (F[j] from F[j-1] and input j, B[j] from B[j+1] and input j, output
 * j from F[j-1] and B[j+1]); the arithmetic differs from the libm expression by a few 1e-16 absolute.
 *
 * E must be a normal number: the caller uses this form only while every input of the node satisfies
 * |v| <= DM_SHARED_LIMIT (partial results never exceed their smaller operand in magnitude) and evaluates the
 * node with dm_boxplus otherwise (e.g. two shortened bits, LLR 99999.9, on one check node).
 * ------------------------------------------------------------------------------------------------ */
#define DM_SHARED_LIMIT 600.0

DM_FN double dm_e_combine(double ex, double ey) { return dm_ratio_div(ex + ey, DM_FMA(ex, ey, 1.0)); } /* E >= e^-600 */

/* The same economy as dm_frac (below) in the E domain: a partial result over m >= 2 inputs is carried as the fraction
   E = n / d and divided only where a message leaves the node,
       first two inputs  a [+] b : n = a + b,           d = 1 + a b
       one more input    f [+] c : n' = f.n + f.d c,    d' = f.d + f.n c
       two partials      f [+] g : (f.n g.d + f.d g.n) / (f.d g.d + f.n g.n)
   so a degree-D node costs D divisions instead of 3(D-2).  E <= 1: n and d at most double per step (no rescaling up to
   the 64 inputs of the shared form), d >= 1, n >= e^-600. */
typedef struct
{
    double n, d;
} dm_efrac;
DM_FN dm_efrac dm_efrac_first(double a, double b)
{
    dm_efrac f;
    f.n = a + b;
    f.d = DM_FMA(a, b, 1.0);
    return f;
}
DM_FN dm_efrac dm_efrac_step(dm_efrac f, double c)
{
    dm_efrac g;
    g.n = DM_FMA(f.d, c, f.n);
    g.d = DM_FMA(f.n, c, f.d);
    return g;
}
DM_FN double dm_efrac_e(dm_efrac f) { return dm_ratio_div(f.n, f.d); }
DM_FN double dm_efrac_e2(dm_efrac f, dm_efrac g)
{
    double num = DM_FMA(f.n, g.d, f.d * g.n);
    double den = DM_FMA(f.n, g.n, f.d * g.d);
    return dm_ratio_div(num, den);
}

/* LLR of a partial result carried as (sign bit, E): s * (-log E); an exact zero comes out as +0.0, as in
   the reference where log(1) = +0.0 is added to a signed zero */
DM_FN double dm_e_to_llr(uint32_t sign_word, double e)
{
    double mag = 0.0 - dm_boxplus_log(e);
    double sd = dm_from_bits((uint64_t)(0x3FF00000u | (sign_word & 0x80000000u)) << 32); /* s as +-1.0 */
    return DM_FMA(sd, mag, 0.0);
}

/* sign carried as the upper word of the double (bit 31 = sign); signs combine by xor of these words */
#define DM_SIGN_WORD(x) ((uint32_t)(dm_bits(x) >> 32))

/* ------------------------------------------------------------------------------------------------
 * Saturated check nodes.  Without early termination a converged frame keeps iterating and its LLRs double with every
 * pass — 1e13 after 50 iterations of the (3,6) code, far beyond DM_SHARED_LIMIT — while the DIFFERENCES between the
 * inputs of one check node stay of the order of the channel LLRs.  With mu = min_j |v_j| and E'_j = e^-(|v_j| - mu)
 * the E-domain box-plus of two partial results (a e^-mu) and (b e^-mu) is  e^-mu (a + b) / (1 + e^-2mu a b).  Once
 * mu >= DM_SAT_MIN = 40 the denominator differs from 1 by less than 64^2 e^-80 < 2^-103, and the recursion is a sum:
 *
 *     |c2v_j| = mu - log( sum_{i != j} E'_i ),        sign as in the E-domain form
 *
 * with the sums taken in the reference's order (F[j] = F[j-1] + E'_j, B[j] = B[j+1] + E'_j, F[j-1] + B[j+1]): d
 * exponentials of small arguments, d logarithms and 3(d-2) additions for a degree-d node, where the chain of
 * dm_boxplus costs 3(d-2) times two exponentials, a division and a logarithm — and it is the value of the reference's
 * chain of box-pluses to within 2^-103 relative, before rounding.
 * Rule (3 <= d <= 64, checked BEFORE the E-domain form): mu >= DM_SAT_MIN and max_j |v_j| - mu <= DM_SHARED_LIMIT,
 * so that every E' is a normal number and every sum lies in [e^-600, 64].
 * ------------------------------------------------------------------------------------------------ */
#define DM_SAT_MIN 40.0
DM_FN int dm_sat_applies(double mu, double amax) { return mu >= DM_SAT_MIN && amax - mu <= DM_SHARED_LIMIT; }
/*
 * Later still the differences double too and leave that window (amax - mu > DM_SHARED_LIMIT): a converged frame of the
 * (3,6) code reaches it around iteration 30 of 50, and the chain of dm_boxplus that was the fall-back costs three times the
 * saturated form (measured on the n = 8192 code: 0.45 ms per iteration of an 8 192-frame batch in the window, 1.27 ms
 * beyond it, where a wave's lanes split between the two and the wave pays for both).  The saturated form needs only ONE
 * more thing there: the output of the edge that HOLDS the minimum mu is the combination of the OTHER inputs, which may all
 * be far away — their E' = e^-(|v| - mu) underflow — so that one output takes its own base: with jm the first index where
 * |v| = mu and m2 = min_{i != jm} |v_i|,
 *
 *     |c2v_jm| = m2 - log( sum_{i != jm} e^-min(|v_i| - m2, 700) )      (in index order; the sum contains a 1)
 *
 * and every other output is the saturated form's, base mu (its sum contains E'_jm = 1; inputs more than 700 above the base
 * enter as e^-700 instead of less: 64 e^-700 relative, nothing).  Same error analysis as above, no condition on the spread.
 * Rule (nodes of degree 5..16: below, the chain is three or six box-pluses that mostly take their +-min short cut, cheaper
 * than this form — measured on h.txt; above 16 the wide-node scratch form keeps the chain): mu >= DM_SAT_MIN and
 * amax - mu > DM_SHARED_LIMIT -> the saturated form with that one output replaced.
 */
DM_FN int dm_sat2_applies(double mu, double amax) { return mu >= DM_SAT_MIN && amax - mu > DM_SHARED_LIMIT; }
DM_FN double dm_sat_e(double absv, double mu) { return dm_boxplus_exp(absv - mu); }
DM_FN double dm_sat_mag(double mu, double sum) { return mu - dm_boxplus_log(sum); }
DM_FN double dm_sat_signed(uint32_t sign_word, double mag)
{
    double sd = dm_from_bits((uint64_t)(0x3FF00000u | (sign_word & 0x80000000u)) << 32); /* s as +-1.0 */
    return DM_FMA(sd, mag, 0.0);
}
DM_FN double dm_sat_llr(uint32_t sign_word, double mu, double sum) { return dm_sat_signed(sign_word, dm_sat_mag(mu, sum)); }

/* ------------------------------------------------------------------------------------------------
 * Likelihood-ratio form of the whole BP iteration (no exp/log inside the loop).
 *
 * With rho(L) = e^L and lambda(L) = e^-L the reference's two updates are, as real functions,
 *   variable node (decoder.cpp:48-64):  lambda(total)  = lambda(L_ch) * prod_p lambda(c2v_p)
 *                                       rho(v2c_p)     = lambda(c2v_p) / lambda(total)
 *   check node    (decoder.cpp:25-45):  rho(x [+] y)    = (1 + rho(x) rho(y)) / (rho(x) + rho(y))
 *                                       lambda(x [+] y) = (rho(x) + rho(y)) / (1 + rho(x) rho(y))
 * so v2c messages are carried as rho, c2v messages as lambda; an iteration costs one IEEE division per
 * c2v message and one per variable node, and every operand is positive: no cancellation, the relative error
 * of a ratio (= the absolute error of its LLR) stays at a few 1e-16 per operation.
 *
 * binary64 holds e^L only for |L| < 709, so this form is valid while every message stays inside
 * [DM_RATIO_LO, DM_RATIO_HI) = 2^-+240 (|L| < 166.4) and every channel LLR within DM_RATIO_LLR_LIMIT; a
 * frame that leaves that box at any point is decoded from scratch with the LLR-domain form above (dm_boxplus /
 * dm_e_combine).  The rule is per frame and depends on nothing but the frame's own data.  Products of up to
 * four in-range factors stay normal numbers (2^-+960); longer products are range-checked every third factor.
 * ------------------------------------------------------------------------------------------------ */
#define DM_RATIO_HI 0x1p240
#define DM_RATIO_LO 0x1p-240
#define DM_RATIO_LLR_LIMIT 166.0

DM_FN int dm_ratio_out_of_range(double r) { return !(r >= DM_RATIO_LO && r < DM_RATIO_HI); } /* NaN: out */
/*
 * The same predicate on the upper word of the bit pattern (positive doubles order like their bit patterns, and both
 * bounds are powers of two, so the lower word never matters): r is outside [2^-240, 2^240) exactly when
 * dm_ratio_key(r) >= DM_RATIO_KEY_SPAN in unsigned arithmetic — zero, denormals, negative numbers, infinities
 * and NaNs included.  The kernels keep the running maximum of the keys (DM_RATIO_TRACK) and compare once per pass.
 */
#define DM_RATIO_KEY_LO ((1023u - 240u) << 20)
#define DM_RATIO_KEY_SPAN (480u << 20)
DM_FN uint32_t dm_ratio_key(double r) { return (uint32_t)(dm_bits(r) >> 32) - DM_RATIO_KEY_LO; }
#define DM_RATIO_TRACK(acc, r) ((acc) = (acc) > dm_ratio_key(r) ? (acc) : dm_ratio_key(r))
#define DM_RATIO_ESCAPED(acc) ((acc) >= DM_RATIO_KEY_SPAN)

/*
 * dm_ratio_div(a, b) = the correctly rounded quotient a / b for the operands this form produces: positive normal
 * numbers with a, b and a/b all inside 2^-+1000 (guaranteed while the frame is inside the box; once a value has
 * left it the frame's results are discarded, so what a division returns after that point does not matter, and the
 * value that trips a range check is always a product, never a quotient).  On the host it is the IEEE division.
 * On the device it is the hardware's own division sequence (reciprocal estimate, two Newton steps, quotient,
 * one correction: what the compiler emits for `/`) without the operand pre-scaling and special-case fix-up, which
 * are the identity for such operands: 8 instructions instead of 11, same bits.
 */
DM_FN double dm_ratio_div(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(b);
    double e = DM_FMA(-b, r, 1.0);
    r = DM_FMA(r, e, r);
    e = DM_FMA(-b, r, 1.0);
    r = DM_FMA(r, e, r);
    double q = a * r;
    double t = DM_FMA(-b, q, a);
    return DM_FMA(t, r, q);
#else
    return a / b;
#endif
}
/*
 * a / b for a divisor that is used many times, given r = the correctly rounded 1 / b (computed once, on the host, by the IEEE
 * division): q = RN(a r) is within one ulp of a / b, the remainder a - b q is exact in one fma, and RN(q + (a - b q) r) is the
 * correctly rounded quotient (Markstein's theorem: it needs r correctly rounded, which a Newton iteration does not promise
 * but the host's division does).  Three instructions where the device's general division takes fifteen; on the host it IS
 * the division.  For finite a, normal b, quotient and remainder far from the ends of the exponent range (the channel's
 * 2 y / sigma^2: |y| < 2^7, sigma^2 in [2^-7, 2^7]); checked against the device's IEEE division by ldpc_hip_selftest_division.
 */
DM_FN double dm_div_by(double a, double b, double r)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double q = a * r;
    double t = DM_FMA(-b, q, a);
    return DM_FMA(t, r, q);
#else
    (void)r;
    return a / b;
#endif
}
DM_FN double dm_ratio_rho(double x, double y) { return dm_ratio_div(DM_FMA(x, y, 1.0), x + y); }
DM_FN double dm_ratio_lambda(double x, double y) { return dm_ratio_div(x + y, DM_FMA(x, y, 1.0)); }
/* the same two with one operand given as a fraction n/d (a partial result not yet divided) */
DM_FN double dm_ratio_lambda_frac(double n, double d, double y) { return dm_ratio_div(DM_FMA(d, y, n), DM_FMA(n, y, d)); }

/*
 * Check nodes of any degree without dividing partial results: a partial result of the forward/backward recursion
 * (decoder.cpp:31-44) over m >= 2 inputs is carried as the fraction rho = n / d,
 *     first two inputs   a [+] b : n = 1 + a b,   d = a + b
 *     one more input     f [+] c : n' = f.d + f.n c,  d' = f.n + f.d c
 * and divided only where a c2v message leaves the node:
 *     lambda(f [+] c) = (f.n + f.d c) / (f.d + f.n c)                              (dm_ratio_lambda_frac)
 *     lambda(f [+] g) = (f.n g.d + f.d g.n) / (f.d g.d + f.n g.n)                  (dm_frac_lambda2)
 * so a degree-D node costs D divisions instead of 3(D-2).  Every operand is positive: no cancellation.
 * Magnitudes: inputs lie in [2^-240, 2^240).  A partial over an ODD number m >= 3 of inputs is rescaled, right
 * after it has been formed, by the power of two that brings n into [1, 2) (dm_frac_norm: exact, it changes no
 * later rounding); then n < 2^721, d < 2^722 before a rescale and n < 2^481, d < 2^482 for every stored partial,
 * and the products of dm_frac_lambda2 stay below 2^964.  From below, n and d never fall under 2^-241.
 */
typedef struct
{
    double n, d;
} dm_frac;

DM_FN dm_frac dm_frac_first(double a, double b)
{
    dm_frac f;
    f.n = DM_FMA(a, b, 1.0);
    f.d = a + b;
    return f;
}
DM_FN dm_frac dm_frac_step(dm_frac f, double c)
{
    dm_frac g;
    g.n = DM_FMA(f.n, c, f.d);
    g.d = DM_FMA(f.d, c, f.n);
    return g;
}
DM_FN dm_frac dm_frac_norm(dm_frac f)
{
    /* s = 2^-(unbiased exponent of n): exponent field of s = 2046 - exponent field of n */
    double s = dm_from_bits(0x7FE0000000000000ull - (dm_bits(f.n) & 0x7FF0000000000000ull));
    dm_frac g;
    g.n = f.n * s;
    g.d = f.d * s;
    return g;
}
DM_FN double dm_frac_lambda2(dm_frac f, dm_frac g)
{
    double num = DM_FMA(f.n, g.d, f.d * g.n);
    double den = DM_FMA(f.n, g.n, f.d * g.d);
    return dm_ratio_div(num, den);
}

/*
 * Shared-reciprocal check nodes (likelihood-ratio form WITH early termination, check nodes of degree 3 and 4).
 *
 * Such a node divides three or four times, each time by a different denominator d_k, and a division is the most
 * expensive thing the iteration does (a 16-cycle reciprocal estimate, two Newton steps, the quotient and its correction).
 * Here a node of degree 3 takes ONE correctly rounded reciprocal r = 1 / P of the product P = d_0 d_1 d_2 and recovers every
 * 1 / d_k from it with multiplications (Montgomery's simultaneous inversion); a node of degree 4 does the same for its
 * denominators two by two:
 *
 *     degree 3    p01 = d0 d1,  P = p01 d2,  r = 1/P
 *                 i2 = r p01,   t = r d2,   i0 = t d1,   i1 = t d0                  lambda(c2v_k) = n_k i_k
 *     degree 4    p01 = d0 d1,  r01 = 1/p01,  i0 = r01 d1,  i1 = r01 d0;   p23 = d2 d3,  r23 = 1/p23,  i2 = r23 d3,  i3 = r23 d2
 *
 * with the numerators n_k and denominators d_k of dm_ratio_lambda (degree 3) and of the fraction form (degree 4: the two
 * partial results F[1], B[2] stay undivided, dm_ratio_lambda_frac) formed exactly as before.  22 instructions and one
 * reciprocal instead of 30 and three (degree 3), 36 and two instead of 44 and four (degree 4).  Every factor is positive
 * and every operation is a plain binary64 multiply, so an output carries about three more roundings than a quotient would:
 * a relative error of a few 1e-16 on a ratio, i.e. that much ABSOLUTE error on the message's LLR per iteration.
 *
 * Range.  Inputs lie in [2^-240, 2^240).  Degree 3: d = 1 + ab in [1, 2^481); degree 4: d in [2^-240, 2^723).  A product
 * never underflows (>= 2^-480) but may exceed the double range when several inputs are large at once; the node therefore
 * returns the upper word of its (larger) product and a frame in which one reached 2^897 is treated like a frame that left
 * the box: decoded again from scratch (separately divided outputs first, then the LLR domain: three launches).  The
 * threshold is chosen so that the check rides on the escape tracking the form has anyway: (upper word of P) >> 2 reaches DM_RATIO_KEY_SPAN exactly when P >= 2^897 — infinities, NaNs and negative values
 * included, positive doubles order like their bit patterns — so DM_SHARED_TRACK feeds the same running maximum as
 * DM_RATIO_TRACK (one shift and one max per node, no register of its own; a per-node fall-back to separate quotients was
 * measured first and costs the headline kernel its fifth resident frame: 12 bytes of scratch at 96 registers).  For
 * P < 2^897 every intermediate lies within 2^-+1000 and dm_ratio_div(1, P) is the correctly rounded reciprocal on both
 * sides.  The rule depends on the frame's own data only.  How often it fires (h.txt, AWGN, det-mode oracle): in none of
 * 20 000 frames at -4 and -2 dB, none of 2 000 at 0 dB, 0.15 % at +2 dB, 1.2 % at +4 dB, most frames at +6 dB (where frames
 * leave the box of the ratio form itself) — far above the waterfall, where a frame takes two or three iterations.  (One
 * reciprocal for all four denominators of a degree-4 node, 32 instructions, was measured first: 0.25 % of the frames at
 * -4 dB overflow it, and a frame decoded again costs its whole latency once more, serialised behind the batch: 0.2 ms.)
 * Frames decoded WITHOUT early termination (hand-over form) keep the separately divided outputs: their messages grow until
 * the hand-over and would overflow P first.
 */
#define DM_SHARED_KEY(p_hi) ((uint32_t)(p_hi) >> 2)
#define DM_SHARED_TRACK(acc, p_hi) ((acc) = (acc) > DM_SHARED_KEY(p_hi) ? (acc) : DM_SHARED_KEY(p_hi))
#define DM_SHARED_OVERFLOW(p_hi) (DM_SHARED_KEY(p_hi) >= DM_RATIO_KEY_SPAN) /* P >= 2^897, or not a positive finite number */

DM_FN uint32_t dm_cn3_shared(double *v) /* v[j] = rho(v2c_j) on entry, lambda(c2v_j) on return; returns the upper word of P */
{
    const double n0 = v[2] + v[1], d0 = DM_FMA(v[2], v[1], 1.0); /* B[1] = B[2] [+] v[1] */
    const double n1 = v[0] + v[2], d1 = DM_FMA(v[0], v[2], 1.0); /* F[0] [+] B[2] */
    const double n2 = v[0] + v[1], d2 = DM_FMA(v[0], v[1], 1.0); /* F[1] = F[0] [+] v[1] */
    const double p01 = d0 * d1, P = p01 * d2;
    const double r = dm_ratio_div(1.0, P);
    const double i2 = r * p01, t = r * d2;
    const double i0 = t * d1, i1 = t * d0;
    v[0] = n0 * i0, v[1] = n1 * i1, v[2] = n2 * i2;
    return (uint32_t)(dm_bits(P) >> 32);
}

DM_FN uint32_t dm_cn4_shared(double *v)
{
    const double nF = DM_FMA(v[0], v[1], 1.0), dF = v[0] + v[1]; /* F[1] = nF / dF */
    const double nB = DM_FMA(v[3], v[2], 1.0), dB = v[3] + v[2]; /* B[2] = nB / dB */
    /* two pairs, a reciprocal each: the product of all four denominators (about the cube of the product of the node's four
       inputs) leaves the double range in one frame in four hundred at -4 dB — frames that then have to be decoded again —
       the product of two does so in none of 60 000.  The pairs one after the other (on the device a scheduling fence
       between them): side by side they cost the headline kernel a register it does not have. */
    const double n0 = DM_FMA(dB, v[1], nB), d0 = DM_FMA(nB, v[1], dB); /* B[1] = B[2] [+] v[1] */
    const double n1 = DM_FMA(dB, v[0], nB), d1 = DM_FMA(nB, v[0], dB); /* F[0] [+] B[2] */
    const double p01 = d0 * d1;
    const double r01 = dm_ratio_div(1.0, p01);
    const double o0 = n0 * (r01 * d1), o1 = n1 * (r01 * d0);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
    const double n2 = DM_FMA(dF, v[3], nF), d2 = DM_FMA(nF, v[3], dF); /* F[1] [+] B[3] */
    const double n3 = DM_FMA(dF, v[2], nF), d3 = DM_FMA(nF, v[2], dF); /* F[2] = F[1] [+] v[2] */
    const double p23 = d2 * d3;
    const double r23 = dm_ratio_div(1.0, p23);
    v[0] = o0, v[1] = o1, v[2] = n2 * (r23 * d3), v[3] = n3 * (r23 * d2);
    const uint32_t h01 = (uint32_t)(dm_bits(p01) >> 32), h23 = (uint32_t)(dm_bits(p23) >> 32);
    return h01 > h23 ? h01 : h23;
}

/*
 * Degree 6 (the (3,6)-regular codes' check node): the six outputs of the fraction form above (dm_frac: same partial results,
 * same rescaling, same numerators n_k and denominators d_k as the separately divided node) in two triples, outputs 0,1,2 and
 * 3,4,5, each triple with ONE reciprocal of the product of its three denominators: 2 reciprocals and 32 further
 * instructions where six divisions take 6 and 42.  Returns the larger upper word of the two products (range rule as above:
 * below 2^897; measured on the n = 8192 code from 1 to 6 dB: at most 2^610 — all six in one product reach 2^1095).
 */
#if defined(__HIP_DEVICE_COMPILE__)
#define DM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define DM_TIE5(a, b, c, d, e) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e))
#else
#define DM_SCHED_FENCE() ((void)0)
#define DM_TIE5(a, b, c, d, e) ((void)0)
#endif
DM_FN uint32_t dm_cn6_shared(double *v)
{
    /* (the fences keep the device compiler from interleaving the stages — of this node and of its neighbours in a register-
       resident kernel — which would need more registers than there are; they change no value) */
    const dm_frac B4 = dm_frac_first(v[5], v[4]);
    const dm_frac B3 = dm_frac_norm(dm_frac_step(B4, v[3]));
    const dm_frac B2 = dm_frac_step(B3, v[2]);
    DM_SCHED_FENCE();
    const double n0 = DM_FMA(B2.d, v[1], B2.n), d0 = DM_FMA(B2.n, v[1], B2.d); /* B[1] = B[2] [+] v[1] */
    const double n1 = DM_FMA(B2.d, v[0], B2.n), d1 = DM_FMA(B2.n, v[0], B2.d); /* F[0] [+] B[2] */
    dm_frac F = dm_frac_first(v[0], v[1]);                                      /* F[1] */
    const double n2 = DM_FMA(F.n, B3.d, F.d * B3.n), d2 = DM_FMA(F.n, B3.n, F.d * B3.d); /* F[1] [+] B[3] */
    F = dm_frac_norm(dm_frac_step(F, v[2])); /* F[2] */
    DM_SCHED_FENCE();
    const double p01 = d0 * d1, P = p01 * d2;
    const double r = dm_ratio_div(1.0, P);
    const double i2 = r * p01, t = r * d2;
    v[0] = n0 * (t * d1), v[1] = n1 * (t * d0), v[2] = n2 * i2;
    DM_TIE5(v[0], v[1], v[2], F.n, F.d); /* the second triple starts when the first has been delivered */
    DM_SCHED_FENCE();
    const double n3 = DM_FMA(F.n, B4.d, F.d * B4.n), d3 = DM_FMA(F.n, B4.n, F.d * B4.d); /* F[2] [+] B[4] */
    F = dm_frac_step(F, v[3]);                                                            /* F[3] */
    const double n4 = DM_FMA(F.d, v[5], F.n), d4 = DM_FMA(F.n, v[5], F.d);               /* F[3] [+] B[5] */
    const double n5 = DM_FMA(F.d, v[4], F.n), d5 = DM_FMA(F.n, v[4], F.d);               /* F[4] = F[3] [+] v[4] */
    DM_SCHED_FENCE();
    const double p34 = d3 * d4, Q = p34 * d5;
    const double s = dm_ratio_div(1.0, Q);
    const double i5 = s * p34, u = s * d5;
    v[3] = n3 * (u * d4), v[4] = n4 * (u * d3), v[5] = n5 * i5;
    DM_SCHED_FENCE();
    const uint32_t hP = (uint32_t)(dm_bits(P) >> 32), hQ = (uint32_t)(dm_bits(Q) >> 32);
    return hP > hQ ? hP : hQ;
}

/* ------------------------------------------------------------------------------------------------
 * Fused form (round 4): the first of the three launches of sum-product WITH early termination, for codes whose check
 * nodes have 2..4 edges and at most one degree-1 neighbour each (fused_rule.h: a property of the code alone; the n = 1024
 * test code is such a code).  Same iteration as the likelihood-ratio form above, with three changes that take work out of
 * it.  Every quantity stays positive, nothing cancels; the det-mode oracle runs the same functions.
 *
 * 1. Order of a check node's inputs.  The node takes its neighbours of degree >= 3 first, then those of degree 2, then its
 *    degree-1 neighbour (a LEAF), each group in row file order.  The box-plus is associative and commutative, so the
 *    forward/backward recursion of decoder.cpp:31-44 over this order yields the same messages up to rounding.
 * 2. A leaf never changes its v2c message: out - c2v = L_ch (decoder.cpp:50-64), i.e. rho_ch.  All the node's other
 *    outputs need from it is that constant, and all the decoder needs from the message c2v = n / d the node would send to
 *    it is the hard decision  out <= 0  <=>  lambda(c2v) >= rho_ch  <=>  n >= rho_ch d  (d > 0): one multiplication and a
 *    comparison instead of a division, a message slot and a variable-node visit.  The check node keeps the decision.
 * 3. A variable node of degree 2 divides only to undo its own inputs: rho(v2c_0) = rho_ch rho(c2v_1).  The check node hands
 *    such a neighbour rho(c2v) = d / n instead of lambda(c2v) = n / d — the same reciprocal sequence with numerator and
 *    denominator exchanged ("flipped" output) — and the variable node multiplies: no division at all.  Its decision is
 *    rho(total) = rho_ch rho(c2v_0) rho(c2v_1) <= 1.
 *
 * Reciprocals are shared as in the shared-reciprocal form: a node of degree 3 inverts the product of the (two or three)
 * denominators of its message outputs, a node of degree 4 those of outputs 0,1 and of outputs 2,3 (a leaf's partner is
 * divided on its own).  Every product inverted — a single denominator counts — must stay below 2^897 (DM_FUSED_P_HI on its
 * upper word; infinities, NaNs and negative patterns compare above it): a frame in which one does not is decoded again by
 * the separately divided ratio form, exactly like a frame that left the box.  With inputs inside [2^-240, 2^240) no product
 * falls below 2^-720.  Functions: v[] = rho(v2c) of the node's inputs in the order of item 1 (a leaf's entry = rho_ch) on
 * entry; on return v[k] = the message for neighbour k — lambda(c2v_k), or rho(c2v_k) where bit k of `flip` is set; the
 * leaf's entry is left alone.  *leaf_bit = its hard decision; *leaf_tot (may be null) = lambda(total) of the leaf, for the
 * LLR output.  Returns the largest upper word among the products inverted.  shared = 0: every output is divided on its own
 * (sum-product WITHOUT early termination, where messages grow until the hand-over below and products of denominators
 * would overflow first): nothing to range-check, 0 is returned.
 * ------------------------------------------------------------------------------------------------ */
#define DM_FUSED_P_HI 0x78000000u /* upper word of 2^897 */
DM_FN uint32_t dm_hi(double x) { return (uint32_t)(dm_bits(x) >> 32); }
DM_FN uint32_t dm_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

DM_FN uint32_t dm_cnf2(double *v, unsigned flip)
{
    const double a = v[0], b = v[1];
    uint32_t h = 0;
    if (flip & 1u)
        v[0] = b; /* rho(c2v_0) = rho(v2c_1) */
    else
        v[0] = dm_ratio_div(1.0, b), h = dm_hi(b);
    if (flip & 2u)
        v[1] = a;
    else
        v[1] = dm_ratio_div(1.0, a), h = dm_umax(h, dm_hi(a));
    return h;
}

DM_FN uint32_t dm_cnf3(double *v, unsigned flip, int leaf, int shared, uint32_t *leaf_bit, double *leaf_tot)
{
    const double n0 = v[2] + v[1], d0 = DM_FMA(v[2], v[1], 1.0); /* B[1] = B[2] [+] v[1] */
    const double n1 = v[0] + v[2], d1 = DM_FMA(v[0], v[2], 1.0); /* F[0] [+] B[2] */
    const double n2 = v[0] + v[1], d2 = DM_FMA(v[0], v[1], 1.0); /* F[1] = F[0] [+] v[1] */
    const double N0 = (flip & 1u) ? d0 : n0, D0 = (flip & 1u) ? n0 : d0;
    const double N1 = (flip & 2u) ? d1 : n1, D1 = (flip & 2u) ? n1 : d1;
    const double N2 = (flip & 4u) ? d2 : n2, D2 = (flip & 4u) ? n2 : d2;
    if (leaf) /* v[2] = rho_ch of the leaf */
    {
        const double t = v[2] * d2;
        *leaf_bit = n2 >= t;
        if (leaf_tot)
            *leaf_tot = dm_ratio_div(n2, t);
    }
    if (!shared) /* every output divided on its own (no early termination: the messages grow until the hand-over) */
    {
        v[0] = dm_ratio_div(N0, D0), v[1] = dm_ratio_div(N1, D1);
        if (!leaf)
            v[2] = dm_ratio_div(N2, D2);
        return 0u;
    }
    if (leaf)
    {
        const double P = D0 * D1;
        const double r = dm_ratio_div(1.0, P);
        const double i0 = r * D1, i1 = r * D0;
        v[0] = N0 * i0, v[1] = N1 * i1;
        return dm_hi(P);
    }
    const double p01 = D0 * D1, P = p01 * D2;
    const double r = dm_ratio_div(1.0, P);
    const double i2 = r * p01, t = r * D2;
    const double i0 = t * D1, i1 = t * D0;
    v[0] = N0 * i0, v[1] = N1 * i1, v[2] = N2 * i2;
    return dm_hi(P);
}

DM_FN uint32_t dm_cnf4(double *v, unsigned flip, int leaf, int shared, uint32_t *leaf_bit, double *leaf_tot)
{
    const double nF = DM_FMA(v[0], v[1], 1.0), dF = v[0] + v[1]; /* F[1] = nF / dF */
    const double nB = DM_FMA(v[3], v[2], 1.0), dB = v[3] + v[2]; /* B[2] = nB / dB */
    const double n0 = DM_FMA(dB, v[1], nB), d0 = DM_FMA(nB, v[1], dB); /* B[1] = B[2] [+] v[1] */
    const double n1 = DM_FMA(dB, v[0], nB), d1 = DM_FMA(nB, v[0], dB); /* F[0] [+] B[2] */
    const double N0 = (flip & 1u) ? d0 : n0, D0 = (flip & 1u) ? n0 : d0;
    const double N1 = (flip & 2u) ? d1 : n1, D1 = (flip & 2u) ? n1 : d1;
    double o0, o1;
    uint32_t h01 = 0u;
    if (shared)
    {
        const double p01 = D0 * D1;
        const double r01 = dm_ratio_div(1.0, p01);
        o0 = N0 * (r01 * D1), o1 = N1 * (r01 * D0);
        h01 = dm_hi(p01);
    }
    else
        o0 = dm_ratio_div(N0, D0), o1 = dm_ratio_div(N1, D1);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0); /* the two halves one after the other: registers (dm_cn4_shared) */
#endif
    const double n2 = DM_FMA(dF, v[3], nF), d2 = DM_FMA(nF, v[3], dF); /* F[1] [+] B[3] */
    const double n3 = DM_FMA(dF, v[2], nF), d3 = DM_FMA(nF, v[2], dF); /* F[2] = F[1] [+] v[2] */
    const double N2 = (flip & 4u) ? d2 : n2, D2 = (flip & 4u) ? n2 : d2;
    if (leaf) /* v[3] = rho_ch of the leaf */
    {
        const double o2 = dm_ratio_div(N2, D2);
        const double t = v[3] * d3;
        v[0] = o0, v[1] = o1, v[2] = o2;
        *leaf_bit = n3 >= t;
        if (leaf_tot)
            *leaf_tot = dm_ratio_div(n3, t);
        return shared ? dm_umax(h01, dm_hi(D2)) : 0u;
    }
    const double N3 = (flip & 8u) ? d3 : n3, D3 = (flip & 8u) ? n3 : d3;
    if (!shared)
    {
        v[0] = o0, v[1] = o1, v[2] = dm_ratio_div(N2, D2), v[3] = dm_ratio_div(N3, D3);
        return 0u;
    }
    const double p23 = D2 * D3;
    const double r23 = dm_ratio_div(1.0, p23);
    v[0] = o0, v[1] = o1, v[2] = N2 * (r23 * D3), v[3] = N3 * (r23 * D2);
    return dm_umax(h01, dm_hi(p23));
}

/* the box [2^-240, 2^240) on the upper word of a positive double (what DM_RATIO_TRACK checks, as two running extremes:
   one v_max3 / v_min3 pair per two values on the device); zero, denormals, negative patterns, infinities and NaNs fall
   outside */
#define DM_BOX_HI_WORD ((1023u + 240u) << 20)
#define DM_BOX_LO_WORD ((1023u - 240u) << 20)
DM_FN int dm_box_escaped(uint32_t hi_max, uint32_t hi_min) { return hi_max >= DM_BOX_HI_WORD || hi_min < DM_BOX_LO_WORD; }

/*
 * Hand-over (sum-product WITHOUT early termination).  With the syndrome check off a frame keeps iterating after it has
 * converged and its LLRs grow until they leave the box the ratio form can hold; decoding such frames in the LLR domain
 * from the start would pay its exp/log cost for every iteration.  Instead every frame starts in the ratio form and is
 * handed over to the LLR-domain form — at an iteration boundary, with its messages converted by one logarithm each —
 * as soon as a variable node's total lambda(total) has left [2^-DM_HANDOVER_EXP, 2^DM_HANDOVER_EXP) (|L| >= 152):
 *
 *     loop pass I:  check-node pass I (ratio form)
 *                   a value has left the outer box (DM_RATIO_ESCAPED)       -> decode the frame again from scratch, LLR domain
 *                   I == iterations                                         -> done
 *                   a total of VN pass I-1 has left the inner box           -> c2v_e = -log(lambda_e) for every edge; the
 *                                                                              LLR-domain form continues with VN pass I
 *                   variable-node pass I (ratio form)
 *
 * The rule depends on the frame's own data only.  dm_handover_key(prod) is the exponent distance from 1, two-sided:
 * prod >= 2^k or prod < 2^-k  <=>  key >= k << 20 (signed compare); anything that is not a positive normal number
 * gives a large key.
 */
#define DM_HANDOVER_EXP 220
DM_FN int32_t dm_handover_key(double prod)
{
    int32_t c = (int32_t)((uint32_t)(dm_bits(prod) >> 32) - 0x3FF00000u);
    return c > ~c ? c : ~c;
}
#define DM_HANDOVER_DUE(key_max) ((key_max) >= (int32_t)(DM_HANDOVER_EXP << 20))

#endif /* LDPC_AMD_DETMATH_H */
