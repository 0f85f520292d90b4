// device_math.hpp — scalar building blocks shared by the HIP kernels.
//
// Everything here restates reference arithmetic so that results are bit-identical to the CPU
// oracle built with ORC_MATH_DET (exp/log from detmath.h, everything else IEEE-754 binary64 in
// the reference's own operation order).  Compile with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>

#include "detmath.h"
#include "plan.hpp"

namespace ldpc_amd
{

// Wave-uniform table reads: a pointer re-typed into the constant address space lets the compiler fetch
// block descriptors and work lists with scalar loads (s_load_*) into SGPRs instead of per-lane vector loads.
template <typename T>
using const_as_ptr = const T __attribute__((address_space(4))) *;

template <typename T>
__device__ __forceinline__ const_as_ptr<T> uniform_table(const T *p)
{
    return (const_as_ptr<T>)(reinterpret_cast<uintptr_t>(p));
}

// block descriptors (plan.hpp) read dword by dword through the scalar path
template <typename B>
__device__ __forceinline__ B load_block2(const B *table, uint32_t i) // {u32, u16, u16}
{
    static_assert(sizeof(B) == 8, "descriptor layout");
    const auto t = uniform_table(reinterpret_cast<const uint32_t *>(table));
    const uint32_t w0 = t[2 * i], w1 = t[2 * i + 1];
    return B{w0, static_cast<uint16_t>(w1 & 0xFFFFu), static_cast<uint16_t>(w1 >> 16)};
}

template <typename B>
__device__ __forceinline__ B load_block3(const B *table, uint32_t i) // {u32, u32, u16, u16}
{
    static_assert(sizeof(B) == 12, "descriptor layout");
    const auto t = uniform_table(reinterpret_cast<const uint32_t *>(table));
    const uint32_t w0 = t[3 * i], w1 = t[3 * i + 1], w2 = t[3 * i + 2];
    return B{w0, w1, static_cast<uint16_t>(w2 & 0xFFFFu), static_cast<uint16_t>(w2 >> 16)};
}

// decoder.h:7-10 — sign(x) = 1 - 2*signbit(x)
__device__ __forceinline__ int sgn(double x) { return 1 - 2 * static_cast<int>(__builtin_signbit(x) != 0); }

// std::min(a, b) == (b < a) ? b : a
__device__ __forceinline__ double std_min(double a, double b) { return (b < a) ? b : a; }

// decoder.h:17-20: sign(x) * sign(y) * min(|x|, |y|).  The two signs multiply to +-1, and a multiplication by +-1.0 does
// nothing to the non-negative minimum but set its sign bit (of a zero as well): the value is min(|x|, |y|) carrying
// signbit(x) xor signbit(y).  Three instructions — v_min_f64 with |.| source modifiers, an xor of the high words, a bit
// field insert — where the expression as written compiles to ten (two sign extractions, an integer product, its
// conversion, a compare, two selects, the multiplication): 6 of the 16 lane-instructions per edge-update of the
// min-sum kernel.  std::min(a, b) = (b < a) ? b : a and v_min_f64 differ on NaNs only, which a message never is; the
// instruction is written out because fmin() as a builtin brings a canonicalising v_max_f64 per operand with it.
__device__ __forceinline__ double box_minsum(double x, double y)
{
    double m;
    asm("v_min_f64 %0, |%1|, |%2|" : "=v"(m) : "v"(x), "v"(y));
    const uint64_t mb = dm_bits(m);
    uint32_t hi;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "s"(0x7FFFFFFFu), "v"(static_cast<uint32_t>(mb >> 32)), "v"(DM_SIGN_WORD(x) ^ DM_SIGN_WORD(y)));
    return dm_from_bits((static_cast<uint64_t>(hi) << 32) | (mb & 0xFFFFFFFFull));
}

// decoder.h:12-15 with the deterministic exp/log pair of detmath.h
__device__ __forceinline__ double box_jacobian(double x, double y) { return dm_boxplus(x, y); }

template <bool MINSUM>
__device__ __forceinline__ double boxplus(double x, double y)
{
    if constexpr (MINSUM)
        return box_minsum(x, y);
    else
        return box_jacobian(x, y);
}

// libstdc++ generate_canonical<double,53>(mt19937_64): one draw, double(u64)/2^64, clamped below 1
__device__ __forceinline__ double canonical(uint64_t w)
{
    double r = static_cast<double>(w) * 0x1p-64;
    return (r >= 1.0) ? 0x1.fffffffffffffp-1 : r;
}

// libstdc++ normal_distribution polar step on one trial (u1, u2): accepted iff 0 < r2 <= 1
struct PolarTrial
{
    double x, y, r2;
    __device__ __forceinline__ bool accepted() const { return !(r2 > 1.0 || r2 == 0.0); }
};

__device__ __forceinline__ PolarTrial polar_trial(uint64_t u1, uint64_t u2)
{
    PolarTrial t;
    t.x = 2.0 * canonical(u1) - 1.0;
    t.y = 2.0 * canonical(u2) - 1.0;
    t.r2 = t.x * t.x + t.y * t.y;
    return t;
}

// mt19937_64 tempering (u=29,d=0x5555..., s=17,b=0x71D67FFFEDA60000, t=37,c=0xFFF7EEE000000000, l=43)
__device__ __forceinline__ uint64_t mt_temper(uint64_t z)
{
    z ^= (z >> 29) & 0x5555555555555555ull;
    z ^= (z << 17) & 0x71D67FFFEDA60000ull;
    z ^= (z << 37) & 0xFFF7EEE000000000ull;
    z ^= (z >> 43);
    return z;
}

// mt19937_64 twist: new x[k] = x[k+156] ^ ((upper(x[k]) | lower(x[k+1])) >> 1) ^ (odd ? A : 0)
__device__ __forceinline__ uint64_t mt_twist(uint64_t xk, uint64_t xk1, uint64_t xm)
{
    uint64_t y = (xk & 0xFFFFFFFF80000000ull) | (xk1 & 0x7FFFFFFFull);
    return xm ^ (y >> 1) ^ ((y & 1) ? 0xB5026F5AA96619E9ull : 0ull);
}

} // namespace ldpc_amd
