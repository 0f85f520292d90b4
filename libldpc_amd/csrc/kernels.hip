// kernels.hip — hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// Hot path of heat1q/libldpc rebuilt for CDNA4: flooding BP (sum-product / min-sum) with the
// channel + LLR initialisation fused into the same launch.  One workgroup (4 waves) decodes one frame;
// tens of thousands of frames per launch.  Two residencies of the per-frame state share one kernel body
// (a third, register-resident, lives in kernels_reg.hip):
//   LDS-resident     (n=1024 test code: 27 KB of messages per frame, input LLRs in registers, five frames per CU)
//   memory-resident  (codes that fit neither LDS nor the register-resident kernel: the frames in flight are
//                     sized to stay inside the 256 MiB Infinity Cache)
// and two algebraically equal forms of the sum-product iteration (detmath.h):
//   likelihood-ratio form (RATIO)  BP with early termination: messages e^L / e^-L, no exp/log in the loop, hard
//                                  decisions in the sign bits of the messages, syndrome fused into the CN pass
//   LLR-domain form                BP without early termination, frames the ratio form hands back, min-sum
//
// Reference semantics restated here (file:line in heat1q/libldpc):
//   decode loop            src/decoding/decoder.cpp:11-78
//   box-plus kernels       src/decoding/decoder.h:7-20
//   syndrome early-term    src/decoding/decoder.h:47-64
//   AWGN channel + LLRs    src/sim/channel.cpp:62-93   (libstdc++ normal_distribution, polar method)
//   BSC channel + LLRs     src/sim/channel.cpp:129-162
//   BEC channel + decoder  src/sim/channel.cpp:199-229, src/decoding/decoder.cpp:91-192
//   encoder                src/sim/channel.cpp:44-60, src/core/sparse.h:163-172
//   bit-error count        src/sim/ldpcsim.cpp:184-188
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "device_channel.hpp"
#include "device_cn.hpp"
#include "device_math.hpp"
#include "kernels.hpp"

#ifndef LDPC_AMD_DECODE_PRIO
#define LDPC_AMD_DECODE_PRIO 3
#endif

namespace ldpc_amd
{

namespace
{

constexpr int kThreads = kDecodeWaves * kWaveSize;

// Per-wave phase timers of the likelihood-ratio loop (CN pass, wait at the vote, VN pass, wait at the second
// barrier), compiled in only with -DLDPC_AMD_PHASE_TRACE (LDPC_AMD_PHASE_TRACE_BUILD=1 python -m libldpc_amd.build);
// read back by tools/phase_probe.py.  They are how the lock-step pairs and the in-place work lists were found.
#ifdef LDPC_AMD_PHASE_TRACE
#define PHASE_TIMERS uint64_t tr_cn = 0, tr_w1 = 0, tr_vn = 0, tr_w2 = 0, tr_t = 0;
#define PHASE_START tr_t = __builtin_amdgcn_s_memtime();
#define PHASE_TICK(acc)                                   \
    {                                                     \
        const uint64_t n_ = __builtin_amdgcn_s_memtime(); \
        acc += n_ - tr_t;                                 \
        tr_t = n_;                                        \
    }
#else
#define PHASE_TIMERS
#define PHASE_START
#define PHASE_TICK(acc)
#endif
constexpr uint8_t kErasure = 'E'; // functions.h:105

// check-node update on the frame's message array: slot(j) = m[j*stride] (decoder.cpp:25-45, device_cn.hpp)
template <int D, bool MINSUM>
__device__ __forceinline__ void cn_update(double *m, int stride)
{
    double v[D];
#pragma unroll
    for (int j = 0; j < D; ++j)
        v[j] = m[j * stride];
    cn_core<D, MINSUM>(v);
#pragma unroll
    for (int j = 0; j < D; ++j)
        m[j * stride] = v[j];
}

// A check node wider than the register tiles (degree > 16, memory-resident decoder only): the forward partial
// results F[j] go through a per-frame scratch array laid out like the message array, the backward ones stay in a
// register; same recursion, same order, same arithmetic as cn_core (E-domain form while every input is within
// DM_SHARED_LIMIT and the node has at most 64 edges — the limit of the oracle's shared form — else the reference
// expression term by term).  decoder.cpp:31-44.
// E-domain form of a wide node (SAT: saturated form, detmath.h): partial results as (sign, fraction n/d of E = e^-|L|), the
// sign rides in n's sign bit (n > 0); F[j] (inputs 0..j) goes to scr[j] (n) and scr2[j] (d) for 1 <= j <= d-3, the
// backward partial stays in registers.  Saturated: E' = e^-(|L| - mu), sums instead of fractions (d is not used).
template <bool SAT>
__device__ __forceinline__ void cn_wide_eform(double *m, double *scr, double *scr2, int stride, int d, double mu)
{
    auto enc = [](uint32_t sw, double e) { return dm_from_bits(dm_bits(e) | (static_cast<uint64_t>(sw & 0x80000000u) << 32)); };
    auto ex = [&](double v) { return SAT ? dm_sat_e(__builtin_fabs(v), mu) : dm_boxplus_exp(__builtin_fabs(v)); };
    auto first = [](double a, double b) {
        if constexpr (SAT)
            return dm_efrac{a + b, 1.0};
        else
            return dm_efrac_first(a, b);
    };
    auto step = [](dm_efrac f, double c) {
        if constexpr (SAT)
            return dm_efrac{f.n + c, 1.0};
        else
            return dm_efrac_step(f, c);
    };
    auto llr1 = [&](uint32_t sw, dm_efrac f) { return SAT ? dm_sat_llr(sw, mu, f.n) : dm_e_to_llr(sw, dm_efrac_e(f)); };
    auto llr2 = [&](uint32_t sw, dm_efrac f, dm_efrac g) {
        return SAT ? dm_sat_llr(sw, mu, f.n + g.n) : dm_e_to_llr(sw, dm_efrac_e2(f, g));
    };
    const double v0 = m[0], v1 = m[stride];
    const double e0 = ex(v0);
    dm_efrac F = first(e0, ex(v1));
    uint32_t sF = DM_SIGN_WORD(v0) ^ DM_SIGN_WORD(v1);
    scr[stride] = enc(sF, F.n), scr2[stride] = F.d;
    for (int j = 2; j <= d - 2; ++j)
    {
        const double v = m[j * stride];
        F = step(F, ex(v));
        sF ^= DM_SIGN_WORD(v);
        if (j <= d - 3)
            scr[j * stride] = enc(sF, F.n), scr2[j * stride] = F.d;
    }
    const double vl = m[(d - 1) * stride], vl2 = m[(d - 2) * stride];
    const double el = ex(vl);
    m[(d - 1) * stride] = llr1(sF, F); // c2v[d-1] = F[d-2]
    // j = d-2: F[d-3] [+] raw input d-1
    {
        const double fn = scr[(d - 3) * stride];
        dm_efrac f3;
        f3.n = __builtin_fabs(fn), f3.d = scr2[(d - 3) * stride];
        m[(d - 2) * stride] = llr1(DM_SIGN_WORD(fn) ^ DM_SIGN_WORD(vl), step(f3, el));
    }
    dm_efrac B = first(el, ex(vl2)); // B[d-2]
    uint32_t sB = DM_SIGN_WORD(vl) ^ DM_SIGN_WORD(vl2);
    for (int j = d - 3; j >= 2; --j)
    {
        const double v = m[j * stride];
        const double fn = scr[(j - 1) * stride]; // F[j-1], j-1 >= 1: a fraction
        dm_efrac f;
        f.n = __builtin_fabs(fn), f.d = scr2[(j - 1) * stride];
        m[j * stride] = llr2(DM_SIGN_WORD(fn) ^ sB, f, B); // F[j-1] [+] B[j+1]
        B = step(B, ex(v));                                 // B[j]
        sB ^= DM_SIGN_WORD(v);
    }
    // here B = B[2], sB its sign; j = 1: raw input 0 [+] B[2]; then B[1] = c2v[0]
    m[stride] = llr1(DM_SIGN_WORD(v0) ^ sB, step(B, e0));
    B = step(B, ex(v1));
    sB ^= DM_SIGN_WORD(v1);
    m[0] = llr1(sB, B);
}

template <bool MINSUM>
__device__ __noinline__ void cn_wide(double *m, double *scr, double *scr2, int stride, int d)
{
    if (!MINSUM && d <= 64)
    {
        double amax = 0.0, mu = __builtin_huge_val();
        for (int j = 0; j < d; ++j)
        {
            amax = __builtin_fmax(amax, __builtin_fabs(m[j * stride]));
            mu = __builtin_fmin(mu, __builtin_fabs(m[j * stride]));
        }
        if (dm_sat_applies(mu, amax))
        {
            cn_wide_eform<true>(m, scr, scr2, stride, d, mu);
            return;
        }
        if (amax <= DM_SHARED_LIMIT)
        {
            cn_wide_eform<false>(m, scr, scr2, stride, d, 0.0);
            return;
        }
    }
    double f = m[0];
    scr[0] = f;
    for (int j = 1; j <= d - 2; ++j)
    {
        f = boxplus<MINSUM>(f, m[j * stride]);
        scr[j * stride] = f;
    }
    double bk = m[(d - 1) * stride];
    m[(d - 1) * stride] = f;
    for (int j = d - 2; j >= 1; --j)
    {
        const double v = m[j * stride];
        m[j * stride] = boxplus<MINSUM>(scr[(j - 1) * stride], bk);
        bk = boxplus<MINSUM>(bk, v);
    }
    m[0] = bk;
}

template <bool MINSUM, int MAXD>
__device__ __forceinline__ void cn_block(double *msg, const CnBlock b, int lane, double *scratch = nullptr, double *scratch2 = nullptr)
{
    if (lane >= b.count)
        return;
    double *m = msg + b.off + lane;
    const int s = b.count;
    if constexpr (MAXD >= 16)
        if (b.degree > 16) // wave-uniform
        {
            cn_wide<MINSUM>(m, scratch + b.off + lane, scratch2 + b.off + lane, s, b.degree);
            return;
        }
    switch (b.degree) // wave-uniform
    {
    case 2: cn_update<2, MINSUM>(m, s); break;
    case 3: cn_update<3, MINSUM>(m, s); break;
    case 4: cn_update<4, MINSUM>(m, s); break;
    default:
        if constexpr (MAXD > 4)
        {
            switch (b.degree)
            {
            case 5: cn_update<5, MINSUM>(m, s); break;
            case 6: cn_update<6, MINSUM>(m, s); break;
            case 7: cn_update<7, MINSUM>(m, s); break;
            case 8: cn_update<8, MINSUM>(m, s); break;
            default: break;
            }
        }
        if constexpr (MAXD > 8)
        {
            switch (b.degree)
            {
            case 9: cn_update<9, MINSUM>(m, s); break;
            case 10: cn_update<10, MINSUM>(m, s); break;
            case 11: cn_update<11, MINSUM>(m, s); break;
            case 12: cn_update<12, MINSUM>(m, s); break;
            case 13: cn_update<13, MINSUM>(m, s); break;
            case 14: cn_update<14, MINSUM>(m, s); break;
            case 15: cn_update<15, MINSUM>(m, s); break;
            case 16: cn_update<16, MINSUM>(m, s); break;
            default: break;
            }
        }
        break;
    }
}

// two full blocks (64 nodes each) at once: one LDS round trip for both (min-sum is bound by exactly that latency)
template <int D0, int D1, bool MINSUM>
__device__ __forceinline__ void cn_update2(double *m0, double *m1)
{
    double v0[D0], v1[D1];
#pragma unroll
    for (int j = 0; j < D0; ++j)
        v0[j] = m0[j * kWaveSize];
#pragma unroll
    for (int j = 0; j < D1; ++j)
        v1[j] = m1[j * kWaveSize];
    cn_core<D0, MINSUM>(v0);
    cn_core<D1, MINSUM>(v1);
#pragma unroll
    for (int j = 0; j < D0; ++j)
        m0[j * kWaveSize] = v0[j];
#pragma unroll
    for (int j = 0; j < D1; ++j)
        m1[j * kWaveSize] = v1[j];
}

// returns false when the two degrees have no paired form (the caller then takes the blocks one after the other)
template <bool MINSUM, int MAXD>
__device__ __forceinline__ bool cn_pair(double *msg, uint32_t off0, uint32_t off1, int deg0, int deg1, int lane)
{
    double *m0 = msg + off0 + lane, *m1 = msg + off1 + lane;
    if (deg0 == deg1)
    {
        switch (deg0) // wave-uniform
        {
        case 2: cn_update2<2, 2, MINSUM>(m0, m1); return true;
        case 3: cn_update2<3, 3, MINSUM>(m0, m1); return true;
        case 4: cn_update2<4, 4, MINSUM>(m0, m1); return true;
        default: break;
        }
        if constexpr (MAXD > 4)
            switch (deg0)
            {
            case 5: cn_update2<5, 5, MINSUM>(m0, m1); return true;
            case 6: cn_update2<6, 6, MINSUM>(m0, m1); return true;
            case 7: cn_update2<7, 7, MINSUM>(m0, m1); return true;
            case 8: cn_update2<8, 8, MINSUM>(m0, m1); return true;
            default: break;
            }
    }
    else if (deg0 == 4 && deg1 == 3) // each wave's list is in descending degree order (plan.cpp)
    {
        cn_update2<4, 3, MINSUM>(m0, m1);
        return true;
    }
    else if (deg0 == 3 && deg1 == 2)
    {
        cn_update2<3, 2, MINSUM>(m0, m1);
        return true;
    }
    return false;
}

// ---- likelihood-ratio form (RATIO instantiations): all messages are positive, so the sign bit of a message slot
// is free and carries the hard decision of the edge's variable node (set by the VN pass, preserved by the CN
// pass).  The CN pass therefore sees the syndrome of the previous iteration for free. ----
__device__ __forceinline__ uint32_t hi_word(double x) { return static_cast<uint32_t>(dm_bits(x) >> 32); }
__device__ __forceinline__ double with_sign(double mag, uint32_t sign_hi) // mag > 0, sign_hi = 0 or 0x80000000
{
    return dm_from_bits(dm_bits(mag) | (static_cast<uint64_t>(sign_hi) << 32));
}

// SH = 1: the shared-reciprocal form of degree-3 and degree-4 nodes (detmath.h); esc = the frame's escape tracking, which the
// nodes' denominator products join.  SH = 2: that of degree-6 nodes (codes the LDS-resident decoder does not take); esc = the
// accumulator the caller merges into the frame's tracking once the frame has gone on to the variable-node pass
template <int D, int SH>
__device__ __forceinline__ uint32_t cn_update_ratio(double *m, int stride, uint32_t &esc)
{
    double v[D];
    uint32_t sg[D], par = 0;
#pragma unroll
    for (int j = 0; j < D; ++j)
    {
        const double x = m[j * stride];
        sg[j] = hi_word(x) & 0x80000000u;
        par ^= sg[j];
        v[j] = __builtin_fabs(x);
    }
    cn_ratio<D, SH == 1, SH == 2>(v, &esc, &esc);
    // (SH = 2: the range check of this pass may be IGNORED — the frame ends at this pass's vote, detmath.h — so what an
    // overflowed node wrote must still carry the decisions: a NaN's own sign bit is cleared before the decision goes in)
#pragma unroll
    for (int j = 0; j < D; ++j)
        m[j * stride] = with_sign(SH == 2 ? __builtin_fabs(v[j]) : v[j], sg[j]);
    return par;
}

template <int MAXD, int SH>
__device__ __forceinline__ uint32_t cn_block_ratio(double *msg, const CnBlock b, int lane, uint32_t &esc);
template <int MAXD, int SH>
__device__ __forceinline__ uint32_t cn_block_ratio_fwd(double *msg, const CnBlock b, int lane, uint32_t &esc)
{
    return cn_block_ratio<MAXD, SH>(msg, b, lane, esc);
}

// two full blocks (64 nodes each) of the same degree at once: two independent chains per lane hide the latency of
// the divisions and of the LDS round trip; the constant stride lets the loads pair up (ds_read2st64_b64)
template <int D0, int D1, int SH>
__device__ __forceinline__ uint32_t cn_update_ratio2(double *m0, double *m1, uint32_t &esc)
{
    double v0[D0], v1[D1];
    uint32_t g0[D0], g1[D1], par0 = 0, par1 = 0;
#pragma unroll
    for (int j = 0; j < D0; ++j)
    {
        const double x0 = m0[j * kWaveSize];
        g0[j] = hi_word(x0) & 0x80000000u;
        par0 ^= g0[j];
        v0[j] = __builtin_fabs(x0);
    }
#pragma unroll
    for (int j = 0; j < D1; ++j)
    {
        const double x1 = m1[j * kWaveSize];
        g1[j] = hi_word(x1) & 0x80000000u;
        par1 ^= g1[j];
        v1[j] = __builtin_fabs(x1);
    }
    cn_ratio<D0, SH == 1, SH == 2>(v0, &esc, &esc);
    cn_ratio<D1, SH == 1, SH == 2>(v1, &esc, &esc);
#pragma unroll
    for (int j = 0; j < D0; ++j)
        m0[j * kWaveSize] = with_sign(SH == 2 ? __builtin_fabs(v0[j]) : v0[j], g0[j]);
#pragma unroll
    for (int j = 0; j < D1; ++j)
        m1[j * kWaveSize] = with_sign(SH == 2 ? __builtin_fabs(v1[j]) : v1[j], g1[j]);
    return par0 | par1;
}

template <int MAXD, int SH>
__device__ __forceinline__ uint32_t cn_pair_ratio(double *msg, uint32_t off0, uint32_t off1, int deg0, int deg1, int lane, uint32_t &esc)
{
    double *m0 = msg + off0 + lane, *m1 = msg + off1 + lane;
    if (deg0 == deg1)
    {
        switch (deg0) // wave-uniform
        {
        case 2: return cn_update_ratio2<2, 2, SH>(m0, m1, esc);
        case 3: return cn_update_ratio2<3, 3, SH>(m0, m1, esc);
        case 4: return cn_update_ratio2<4, 4, SH>(m0, m1, esc);
        default: break;
        }
        if constexpr (MAXD > 4)
            switch (deg0)
            {
            case 5: return cn_update_ratio2<5, 5, SH>(m0, m1, esc);
            case 6: return cn_update_ratio2<6, 6, SH>(m0, m1, esc);
            case 7: return cn_update_ratio2<7, 7, SH>(m0, m1, esc);
            case 8: return cn_update_ratio2<8, 8, SH>(m0, m1, esc);
            default: break;
            }
    }
    else if (deg0 == 4 && deg1 == 3) // each wave's list is in descending degree order (plan.cpp): every separate
        return cn_update_ratio2<4, 3, SH>(m0, m1, esc); // chain costs the wave an LDS + division latency (~400 cycles)
    else if (deg0 == 3 && deg1 == 2)
        return cn_update_ratio2<3, 2, SH>(m0, m1, esc);
    // anything else: one after the other
    return cn_block_ratio_fwd<MAXD, SH>(msg, CnBlock{off0, kWaveSize, static_cast<uint16_t>(deg0)}, lane, esc) |
           cn_block_ratio_fwd<MAXD, SH>(msg, CnBlock{off1, kWaveSize, static_cast<uint16_t>(deg1)}, lane, esc);
}

// returns the parity (bit 31) of the hard decisions on this lane's check node
template <int MAXD, int SH>
__device__ __forceinline__ uint32_t cn_block_ratio(double *msg, const CnBlock b, int lane, uint32_t &esc)
{
    if (lane >= b.count)
        return 0;
    double *m = msg + b.off + lane;
    const int s = b.count;
    switch (b.degree) // wave-uniform
    {
    case 2: return cn_update_ratio<2, SH>(m, s, esc);
    case 3: return cn_update_ratio<3, SH>(m, s, esc);
    case 4: return cn_update_ratio<4, SH>(m, s, esc);
    default: break;
    }
    if constexpr (MAXD > 4)
        switch (b.degree)
        {
        case 5: return cn_update_ratio<5, SH>(m, s, esc);
        case 6: return cn_update_ratio<6, SH>(m, s, esc);
        case 7: return cn_update_ratio<7, SH>(m, s, esc);
        case 8: return cn_update_ratio<8, SH>(m, s, esc);
        default: break;
        }
    if constexpr (MAXD > 8)
        switch (b.degree)
        {
        case 9: return cn_update_ratio<9, SH>(m, s, esc);
        case 10: return cn_update_ratio<10, SH>(m, s, esc);
        case 11: return cn_update_ratio<11, SH>(m, s, esc);
        case 12: return cn_update_ratio<12, SH>(m, s, esc);
        case 13: return cn_update_ratio<13, SH>(m, s, esc);
        case 14: return cn_update_ratio<14, SH>(m, s, esc);
        case 15: return cn_update_ratio<15, SH>(m, s, esc);
        case 16: return cn_update_ratio<16, SH>(m, s, esc);
        default: break;
        }
    return 0;
}

// VN update of one node in likelihood-ratio form, fully unrolled for degree DV (all loads in flight at once):
// lambda(total) = lam * prod_p lambda(c2v_p) in column file order; v2c_p = lambda(c2v_p) / lambda(total).
// Returns lambda(total); the hard decision (total LLR <= 0) goes into the sign bit of every v2c written.
// A degree-1 node (a leaf) is special: out - c2v = L_ch, so its v2c message is the channel ratio rho_ch itself — no
// division, nothing to range-check; `lam` is then rho_ch, the decision total <= 0 is taken as lambda(c2v) >= rho_ch, and
// the value returned (for the LLR output only) is lambda(total) = lambda(c2v) / rho_ch.
// A 16-bit field of a register-held slot word addresses a message: as its element index or (BS, "byte slots": the
// instantiations compiled for small codes, 8 * nnz < 2^16) as its byte offset — the field IS the address then, one
// instruction per message less (the array's own offset rides in the instruction).
template <bool BS>
__device__ __forceinline__ double *slot_ptr(double *msg, uint32_t field)
{
    if constexpr (BS)
        return reinterpret_cast<double *>(reinterpret_cast<char *>(msg) + field);
    else
        return msg + field;
}

template <bool BS = false>
__device__ __forceinline__ double vn_leaf_ratio(double *msg, uint32_t slot, double rho)
{
    double *m = slot_ptr<BS>(msg, slot);
    const double c = __builtin_fabs(*m);
    *m = with_sign(rho, c >= rho ? 0x80000000u : 0u);
    return dm_ratio_div(c, rho); // dead code unless the caller wants the LLR
}

template <int DV>
__device__ __forceinline__ double vn_update_ratio(double *msg, const uint32_t *idx, int count, double lam, uint32_t &escaped)
{
    if constexpr (DV == 1)
        return vn_leaf_ratio(msg, idx[0], lam);
    uint32_t s[DV];
    double c[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        s[p] = idx[p * count];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c[p] = __builtin_fabs(msg[s[p]]);
    double prod = lam;
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        prod *= c[p];
        if (DV > 3 && p % 3 == 2)
            DM_RATIO_TRACK(escaped, prod);
    }
    const uint32_t sign = prod >= 1.0 ? 0x80000000u : 0u; // total LLR <= 0: hard decision 1
    const double tot = dm_ratio_div(1.0, prod);                        // rho(total)
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        const double o = tot * c[p]; // rho(total - c2v_p)
        DM_RATIO_TRACK(escaped, o);
        msg[s[p]] = with_sign(o, sign);
    }
    return prod;
}

// the same update for nodes of degree DV <= 2 whose slot indices the lane keeps in a register (two u16, kLlrRegs),
// one node or two independent nodes (of two blocks) in lock step: no table load in front of the LDS round trip,
// two chains in flight
// (the packed word is made opaque every iteration: it is what stays live — unpacked indices hoisted out of the decode loop
// would occupy registers for the whole kernel.  Making the caller's word opaque in place, without the copy, was tried: one
// move less per block, and the headline kernel's allocation tips over, 20 bytes of scratch.)
template <int DV, bool BS>
__device__ __forceinline__ double vn_small_ratio(double *msg, uint32_t packed, double lam, uint32_t &escaped)
{
    static_assert(DV >= 1 && DV <= 2, "register-held slot indices");
    asm volatile("" : "+v"(packed));
    const uint32_t sl[2] = {packed & 0xFFFFu, packed >> 16};
    if constexpr (DV == 1)
        return vn_leaf_ratio<BS>(msg, sl[0], lam);
    double c[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c[p] = __builtin_fabs(*slot_ptr<BS>(msg, sl[p]));
    double prod = lam;
#pragma unroll
    for (int p = 0; p < DV; ++p)
        prod *= c[p];
    const uint32_t sign = prod >= 1.0 ? 0x80000000u : 0u;
    const double tot = dm_ratio_div(1.0, prod);
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        const double o = tot * c[p];
        DM_RATIO_TRACK(escaped, o);
        *slot_ptr<BS>(msg, sl[p]) = with_sign(o, sign);
    }
    return prod;
}

template <int DV, bool BS>
__device__ __forceinline__ void vn_small_ratio2(double *msg, uint32_t packed_a, uint32_t packed_b, double la, double lb,
                                                uint32_t &escaped, double &pa, double &pb)
{
    static_assert(DV >= 1 && DV <= 2, "register-held slot indices");
    asm volatile("" : "+v"(packed_a), "+v"(packed_b)); // unpack here, every iteration
    const uint32_t sa[2] = {packed_a & 0xFFFFu, packed_a >> 16}, sb[2] = {packed_b & 0xFFFFu, packed_b >> 16};
    double ca[DV], cb[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        ca[p] = __builtin_fabs(*slot_ptr<BS>(msg, sa[p])), cb[p] = __builtin_fabs(*slot_ptr<BS>(msg, sb[p]));
    if constexpr (DV == 1) // two leaves: la, lb are their channel ratios (vn_leaf_ratio)
    {
        *slot_ptr<BS>(msg, sa[0]) = with_sign(la, ca[0] >= la ? 0x80000000u : 0u);
        *slot_ptr<BS>(msg, sb[0]) = with_sign(lb, cb[0] >= lb ? 0x80000000u : 0u);
        pa = dm_ratio_div(ca[0], la), pb = dm_ratio_div(cb[0], lb); // dead code unless the caller wants the LLRs
        return;
    }
    pa = la, pb = lb;
#pragma unroll
    for (int p = 0; p < DV; ++p)
        pa *= ca[p], pb *= cb[p];
    const uint32_t sga = pa >= 1.0 ? 0x80000000u : 0u, sgb = pb >= 1.0 ? 0x80000000u : 0u;
    const double ta = dm_ratio_div(1.0, pa), tb = dm_ratio_div(1.0, pb);
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        const double oa = ta * ca[p], ob = tb * cb[p];
        DM_RATIO_TRACK(escaped, oa), DM_RATIO_TRACK(escaped, ob);
        *slot_ptr<BS>(msg, sa[p]) = with_sign(oa, sga);
        *slot_ptr<BS>(msg, sb[p]) = with_sign(ob, sgb);
    }
}

// A node of up to 16 edges whose slot indices the lane keeps in registers (16 x u16 in eight words, kLlrRegs: the
// first — widest — VN block of every wave): no table load at all, every message read once.
// TWICE: the words are unpacked once for the reads and once more for the writes — two instructions per edge against the
// sixteen registers the unpacked addresses occupy across the product and the division otherwise (the hand-over kernel).
template <int DV, bool TWICE, bool BS>
__device__ __forceinline__ double vn_update_ratio_regs(double *msg, uint32_t (&packed)[8], double lam, uint32_t &escaped)
{
    // the words stay packed across iterations: without the barrier the compiler hoists all 16 unpacked indices out
    // of the decode loop and keeps them live for the whole kernel
    uint32_t pk[(DV + 1) / 2];
    auto opaque = [&] {
#pragma unroll
        for (int i = 0; i < (DV + 1) / 2; ++i)
        {
            pk[i] = packed[i];
            asm volatile("" : "+v"(pk[i]));
        }
    };
    opaque();
    auto slot = [&](int p) { return slot_ptr<BS>(msg, (p & 1) ? pk[p >> 1] >> 16 : pk[p >> 1] & 0xFFFFu); };
    if constexpr (DV == 1)
        return vn_leaf_ratio<BS>(msg, pk[0] & 0xFFFFu, lam);
    double c[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c[p] = __builtin_fabs(*slot(p));
    double prod = lam;
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        prod *= c[p];
        if (DV > 3 && p % 3 == 2)
            DM_RATIO_TRACK(escaped, prod);
    }
    const uint32_t sign = prod >= 1.0 ? 0x80000000u : 0u;
    const double tot = dm_ratio_div(1.0, prod);
    if constexpr (TWICE)
        opaque();
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        const double o = tot * c[p];
        DM_RATIO_TRACK(escaped, o);
        *slot(p) = with_sign(o, sign);
    }
    return prod;
}

// (the instantiations for small codes, BS, take such nodes up to degree 15: the sixteenth message is the two registers the
// headline kernel does not have — its callers send a node of degree 16 through the slot table, kWideMax)
template <bool TWICE, bool BS>
__device__ __forceinline__ double vn_block_ratio_regs(double *msg, uint32_t (&packed)[8], int degree, double lam, uint32_t &escaped)
{
    switch (degree) // wave-uniform, 1..16
    {
#define LDPC_VN(D) \
    case D: return vn_update_ratio_regs<D, TWICE, BS>(msg, packed, lam, escaped);
        LDPC_VN(1) LDPC_VN(2) LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7) LDPC_VN(8)
        LDPC_VN(9) LDPC_VN(10) LDPC_VN(11) LDPC_VN(12) LDPC_VN(13) LDPC_VN(14) LDPC_VN(15)
#undef LDPC_VN
    case 16:
        if constexpr (!BS)
            return vn_update_ratio_regs<16, TWICE, BS>(msg, packed, lam, escaped);
        return lam;
    default: return lam;
    }
}

// pieces of the same update for wider nodes still, N messages at a time: pass 1 multiplies N factors into the
// running product (range check at every third position of the node, as everywhere), pass 2 writes N messages
template <int N>
__device__ __forceinline__ void vn_ratio_pass1(const double *msg, const uint32_t *idx, int count, int p0, double &prod,
                                               uint32_t &escaped)
{
    uint32_t s[N];
    double c[N];
#pragma unroll
    for (int p = 0; p < N; ++p)
        s[p] = idx[(p0 + p) * count];
#pragma unroll
    for (int p = 0; p < N; ++p)
        c[p] = __builtin_fabs(msg[s[p]]);
    const int ph = p0 % 3; // wave-uniform
#pragma unroll
    for (int p = 0; p < N; ++p)
    {
        prod *= c[p];
        if ((ph + p) % 3 == 2)
            DM_RATIO_TRACK(escaped, prod);
    }
}

template <int N>
__device__ __forceinline__ void vn_ratio_pass2(double *msg, const uint32_t *idx, int count, int p0, double tot,
                                               uint32_t sign, uint32_t &escaped)
{
    uint32_t s[N];
    double c[N];
#pragma unroll
    for (int p = 0; p < N; ++p)
        s[p] = idx[(p0 + p) * count];
#pragma unroll
    for (int p = 0; p < N; ++p)
        c[p] = __builtin_fabs(msg[s[p]]);
#pragma unroll
    for (int p = 0; p < N; ++p)
    {
        const double o = tot * c[p];
        DM_RATIO_TRACK(escaped, o);
        msg[s[p]] = with_sign(o, sign);
    }
}

__device__ __forceinline__ double vn_block_ratio(double *msg, const uint32_t *idx, int count, int degree, double lam,
                                                 uint32_t &escaped)
{
    if (degree == 1) // a leaf takes its channel ratio rho_ch = 1 / lambda_ch (here, per call: the slow paths only)
        return vn_update_ratio<1>(msg, idx, count, dm_ratio_div(1.0, lam), escaped);
    switch (degree) // wave-uniform
    {
#define LDPC_VN(D) \
    case D: return vn_update_ratio<D>(msg, idx, count, lam, escaped);
        LDPC_VN(2) LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7) LDPC_VN(8)
#undef LDPC_VN
    default: break;
    }
    // wider nodes still: eight messages at a time, every message read twice (registers stay bounded)
    double prod = lam;
    int p0 = 0;
    for (; p0 + 8 <= degree; p0 += 8)
        vn_ratio_pass1<8>(msg, idx, count, p0, prod, escaped);
    switch (degree - p0)
    {
#define LDPC_VN(R) \
    case R: vn_ratio_pass1<R>(msg, idx, count, p0, prod, escaped); break;
        LDPC_VN(1) LDPC_VN(2) LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7)
#undef LDPC_VN
    default: break;
    }
    const uint32_t sign = prod >= 1.0 ? 0x80000000u : 0u;
    const double tot = dm_ratio_div(1.0, prod);
    for (p0 = 0; p0 + 8 <= degree; p0 += 8)
        vn_ratio_pass2<8>(msg, idx, count, p0, tot, sign, escaped);
    switch (degree - p0)
    {
#define LDPC_VN(R) \
    case R: vn_ratio_pass2<R>(msg, idx, count, p0, tot, sign, escaped); break;
        LDPC_VN(1) LDPC_VN(2) LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7)
#undef LDPC_VN
    default: break;
    }
    return prod;
}

// VN update of one node in the LLR domain (decoder.cpp:48-64), unrolled for degree DV: every message is read once.
// Returns the APP LLR; v2c_p = APP - c2v_p, the hard decision goes to the hard-bit array.
template <int DV>
__device__ __forceinline__ double vn_update_llr(double *msg, uint8_t *hb, const uint32_t *idx, int count, double L, bool store_hb)
{
    uint32_t s[DV];
    double c[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        s[p] = idx[p * count];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c[p] = msg[s[p]];
    double out = L;
#pragma unroll
    for (int p = 0; p < DV; ++p) // sequential sum in column file order
        out += c[p];
    const uint8_t bit = out <= 0;
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        msg[s[p]] = out - c[p];
        if (store_hb) // (false: the decisions of this pass are not needed, see the loop)
            hb[s[p]] = bit;
    }
    return out;
}

// two nodes (of two full blocks) of the same degree DV <= 2 in lock step
template <int DV>
__device__ __forceinline__ void vn_update_llr2(bool store_hb, double *msg, uint8_t *hb, const uint32_t *idx0, const uint32_t *idx1, double L0,
                                               double L1, double &out0, double &out1)
{
    uint32_t s0[DV], s1[DV];
    double c0[DV], c1[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        s0[p] = idx0[p * kWaveSize], s1[p] = idx1[p * kWaveSize];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c0[p] = msg[s0[p]], c1[p] = msg[s1[p]];
    out0 = L0, out1 = L1;
#pragma unroll
    for (int p = 0; p < DV; ++p) // sequential sum in column file order
        out0 += c0[p], out1 += c1[p];
    const uint8_t bit0 = out0 <= 0, bit1 = out1 <= 0;
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        msg[s0[p]] = out0 - c0[p];
        msg[s1[p]] = out1 - c1[p];
        if (store_hb)
        {
            hb[s0[p]] = bit0;
            hb[s1[p]] = bit1;
        }
    }
}

// The same updates with the slot indices in registers (kLlrRegs, min-sum: two u16 per word for nodes of degree <= 2, 16 x u16
// in eight words for the wave's first block): no table load between the barrier and the first message read.  These run
// the passes whose hard decisions nobody reads (no early termination: all but the last) and store none.
template <int DV, bool BS>
__device__ __forceinline__ void vn_small_llr2(double *msg, uint32_t packed_a, uint32_t packed_b, double L0, double L1, double &out0,
                                              double &out1)
{
    static_assert(DV >= 1 && DV <= 2, "register-held slot indices");
    asm volatile("" : "+v"(packed_a), "+v"(packed_b)); // unpack here, every iteration
    const uint32_t s0[2] = {packed_a & 0xFFFFu, packed_a >> 16}, s1[2] = {packed_b & 0xFFFFu, packed_b >> 16};
    double c0[DV], c1[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c0[p] = *slot_ptr<BS>(msg, s0[p]), c1[p] = *slot_ptr<BS>(msg, s1[p]);
    out0 = L0, out1 = L1;
#pragma unroll
    for (int p = 0; p < DV; ++p) // sequential sum in column file order
        out0 += c0[p], out1 += c1[p];
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        *slot_ptr<BS>(msg, s0[p]) = out0 - c0[p];
        *slot_ptr<BS>(msg, s1[p]) = out1 - c1[p];
    }
}

template <int DV, bool BS>
__device__ __forceinline__ double vn_small_llr(double *msg, uint32_t packed, double L)
{
    static_assert(DV >= 1 && DV <= 2, "register-held slot indices");
    asm volatile("" : "+v"(packed));
    const uint32_t sl[2] = {packed & 0xFFFFu, packed >> 16};
    double c[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c[p] = *slot_ptr<BS>(msg, sl[p]);
    double out = L;
#pragma unroll
    for (int p = 0; p < DV; ++p)
        out += c[p];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        *slot_ptr<BS>(msg, sl[p]) = out - c[p];
    return out;
}

template <int DV, bool BS>
__device__ __forceinline__ double vn_update_llr_regs(double *msg, uint32_t (&packed)[8], double L)
{
    // the words stay packed across iterations (see vn_update_ratio_regs) and are unpacked once for the reads and once
    // more for the writes: sixteen unpacked addresses kept across the sum are sixteen registers
    uint32_t pk[(DV + 1) / 2];
    auto opaque = [&] {
#pragma unroll
        for (int i = 0; i < (DV + 1) / 2; ++i)
        {
            pk[i] = packed[i];
            asm volatile("" : "+v"(pk[i]));
        }
    };
    auto slot = [&](int p) { return slot_ptr<BS>(msg, (p & 1) ? pk[p >> 1] >> 16 : pk[p >> 1] & 0xFFFFu); };
    opaque();
    double c[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c[p] = *slot(p);
    double out = L;
#pragma unroll
    for (int p = 0; p < DV; ++p) // sequential sum in column file order
        out += c[p];
    opaque();
#pragma unroll
    for (int p = 0; p < DV; ++p)
        *slot(p) = out - c[p];
    return out;
}

template <bool BS>
__device__ __forceinline__ double vn_block_llr_regs(double *msg, uint32_t (&packed)[8], int degree, double L)
{
    switch (degree) // wave-uniform, 1..16
    {
#define LDPC_VN(D) \
    case D: return vn_update_llr_regs<D, BS>(msg, packed, L);
        LDPC_VN(1) LDPC_VN(2) LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7) LDPC_VN(8)
        LDPC_VN(9) LDPC_VN(10) LDPC_VN(11) LDPC_VN(12) LDPC_VN(13) LDPC_VN(14) LDPC_VN(15) LDPC_VN(16)
#undef LDPC_VN
    default: return L;
    }
}

__device__ __forceinline__ double vn_block_llr(double *msg, uint8_t *hb, const uint32_t *idx, int count, int degree, double L, bool store_hb)
{
    switch (degree) // wave-uniform
    {
#define LDPC_VN(D) \
    case D: return vn_update_llr<D>(msg, hb, idx, count, L, store_hb);
        LDPC_VN(1) LDPC_VN(2) LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7) LDPC_VN(8)
        LDPC_VN(9) LDPC_VN(10) LDPC_VN(11) LDPC_VN(12) LDPC_VN(13) LDPC_VN(14) LDPC_VN(15) LDPC_VN(16)
#undef LDPC_VN
    default: break;
    }
    double out = L;
    for (int p = 0; p < degree; ++p)
        out += msg[idx[p * count]];
    const uint8_t bit = out <= 0;
    for (int p = 0; p < degree; ++p)
    {
        const uint32_t sl = idx[p * count];
        msg[sl] = out - msg[sl];
        if (store_hb)
            hb[sl] = bit;
    }
    return out;
}

// the same for the register-budgeted kernels: up to eight edges unrolled, wider nodes eight messages at a time with every
// message read twice (the sum still runs in column file order)
template <int N>
__device__ __forceinline__ void vn_llr_pass1(const double *msg, const uint32_t *idx, int count, int p0, double &out)
{
    uint32_t s[N];
    double c[N];
#pragma unroll
    for (int p = 0; p < N; ++p)
        s[p] = idx[(p0 + p) * count];
#pragma unroll
    for (int p = 0; p < N; ++p)
        c[p] = msg[s[p]];
#pragma unroll
    for (int p = 0; p < N; ++p)
        out += c[p];
}

template <int N>
__device__ __forceinline__ void vn_llr_pass2(double *msg, uint8_t *hb, const uint32_t *idx, int count, int p0, double out, bool store_hb)
{
    uint32_t s[N];
    double c[N];
#pragma unroll
    for (int p = 0; p < N; ++p)
        s[p] = idx[(p0 + p) * count];
#pragma unroll
    for (int p = 0; p < N; ++p)
        c[p] = msg[s[p]];
    const uint8_t bit = out <= 0;
#pragma unroll
    for (int p = 0; p < N; ++p)
    {
        msg[s[p]] = out - c[p];
        if (store_hb)
            hb[s[p]] = bit;
    }
}

__device__ __forceinline__ double vn_block_llr_lean(double *msg, uint8_t *hb, const uint32_t *idx, int count, int degree, double L, bool store_hb)
{
    switch (degree) // wave-uniform
    {
#define LDPC_VN(D) \
    case D: return vn_update_llr<D>(msg, hb, idx, count, L, store_hb);
        LDPC_VN(1) LDPC_VN(2) LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7) LDPC_VN(8)
#undef LDPC_VN
    default: break;
    }
    double out = L;
    int p0 = 0;
    for (; p0 + 8 <= degree; p0 += 8)
        vn_llr_pass1<8>(msg, idx, count, p0, out);
    for (int p = p0; p < degree; ++p)
        out += msg[idx[p * count]];
    for (p0 = 0; p0 + 8 <= degree; p0 += 8)
        vn_llr_pass2<8>(msg, hb, idx, count, p0, out, store_hb);
    const uint8_t bit = out <= 0;
    for (int p = p0; p < degree; ++p)
    {
        const uint32_t sl = idx[p * count];
        msg[sl] = out - msg[sl];
        if (store_hb)
            hb[sl] = bit;
    }
    return out;
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// BP / min-sum decoder, one workgroup (4 waves) per frame.
//   per-frame state: msg[nnz] f64 (v2c and c2v share the slot: each edge is rewritten by its own CN
//   lane, then by its own VN lane), llr[nc] f64 (VN rank order), hb[nnz] u8 (hard decision of the
//   edge's VN, read back by the CN lanes for the syndrome).
// ---------------------------------------------------------------------------------------------
// MAXD = widest check node the instantiation handles (4, 8 or 16): the register allocation of a kernel is
// that of its widest CN update, so narrow codes get a leaner kernel.
// LLR_MODE = where the input LLRs (read once per VN per iteration) live: kLlrLds, kLlrMem (device memory), or
// kLlrRegs: every lane keeps the LLRs of the <= kMaxVnBlocksInRegs variable nodes it serves in registers — VN
// blocks are dealt to waves once per code, so lane (wave, l) always serves the same nodes.  Without the LLR
// array the n=1024 code needs 31 KB of LDS per frame instead of 40 KB: five resident frames per CU, not four.
enum : int { kLlrLds = 0, kLlrMem = 1, kLlrRegs = 2 };
constexpr int kMaxVnBlocksInRegs = 8;

// RATIO = the likelihood-ratio form of the sum-product iteration (detmath.h): v2c messages are rho = e^L, c2v
// messages lambda = e^-L, the input LLRs are kept as lambda.  A frame whose values leave the representable box
// is not finished here: its index goes to a.redo_list and the LLR-domain instantiation decodes it from scratch
// in a second launch (a.redo_list_in / a.redo_count_in).
// HANDOVER (RATIO only, used without early termination): a frame whose variable-node totals approach the edge of the
// box is handed to the LLR-domain instantiation at an iteration boundary, its messages converted by one logarithm each
// (detmath.h "Hand-over"): it is appended to a.redo_list with the iteration to resume at (a.redo_iter) and its c2v
// messages go to a.ws_handover.  The LLR-domain instantiation resumes such frames (a.redo_iter_in / a.ws_handover).
// SEPARATE (RATIO, LDS-resident, early termination): the frames that escaped from the shared-reciprocal form are decoded
// again from scratch with every check-node output divided separately (second of three launches, detmath.h).
// VNB (kLlrRegs) = VN blocks per wave the instantiation provides registers for: every one of them costs three registers
// (LLR, packed slot indices) whether the code has that many blocks or not, so the kernels pinned at five waves per SIMD
// are compiled for the five blocks per wave of an n = 1024 code and codes with more take the general instantiation.
// CHAIN (decode_kernel_list): the frame is entry `slot` of a.redo_list_in (the caller has checked the count); a frame that leaves
// the ratio form's range is not appended to a list: the function returns true and the caller goes on to the next form.
template <bool MINSUM, bool WANT_LLR, bool LDS_RESIDENT, int MAXD, int LLR_MODE, bool RATIO, bool HANDOVER = false, bool SEPARATE = false,
          int VNB = kMaxVnBlocksInRegs, bool CHAIN = false>
__device__ __forceinline__ bool decode_body(const DecodeArgs &a, const uint32_t slot)
{
    static_assert(VNB >= 1 && VNB <= kMaxVnBlocksInRegs, "register-held VN blocks");
    static_assert(!(RATIO && MINSUM), "the ratio form is a sum-product form");
    static_assert(!HANDOVER || RATIO, "the hand-over leaves the ratio form");
    extern __shared__ double lds[];
    __shared__ int misc[4];
    __shared__ int votes[2][kDecodeWaves];
    const DevPlan &P = a.plan;
    const int nnz = P.nnz, nc = P.nc;
    uint64_t frame = slot;
    uint32_t resume_at = 0xFFFFFFFFu; // LLR-domain second pass: iteration a handed-over frame resumes at (else: from scratch)
    if (a.redo_count_in) // second pass: only the frames the ratio form handed back
    {
        if constexpr (!CHAIN)
            if (slot >= *uniform_table(a.redo_count_in))
                return false;
        frame = uniform_table(a.redo_list_in)[slot];
        if (a.redo_iter_in)
            resume_at = uniform_table(a.redo_iter_in)[slot];
    }
    [[maybe_unused]] const bool resuming = !RATIO && resume_at != 0xFFFFFFFFu;
    double *msg, *llr;
    uint8_t *hb;
    if constexpr (LDS_RESIDENT)
    {
        msg = lds;
        if constexpr (LLR_MODE == kLlrLds)
        {
            llr = lds + nnz;
            hb = reinterpret_cast<uint8_t *>(llr + nc);
        }
        else
        {
            // kLlrRegs: the channel writes the LLRs into the (still unused) message array, from where each lane
            // picks up its own before the v2c initialisation overwrites it (nc <= nnz: no isolated VN)
            llr = LLR_MODE == kLlrMem ? a.ws_llr + frame * nc : lds;
            hb = reinterpret_cast<uint8_t *>(lds + nnz);
        }
    }
    else
    {
        msg = a.ws_msg + frame * nnz;
        llr = a.ws_llr + frame * nc;
        hb = a.ws_hb + frame * nnz;
    }
    [[maybe_unused]] double *scratch = (!LDS_RESIDENT && a.ws_scr) ? a.ws_scr + frame * 2 * nnz : nullptr; // cn_wide
    [[maybe_unused]] double *scratch2 = scratch ? scratch + nnz : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform: block descriptors load as scalars
#ifdef LDPC_AMD_PHASE_TRACE
    const uint64_t tr_entry = __builtin_amdgcn_s_memtime();
#endif
    // the decode waves win the SIMD's instruction arbitration over the slab kernel's waves of the next batch's noise, which
    // share the compute units with them, are many and are in no hurry (rng_kernels.hip: the generator's serial chains run at
    // this priority too)
    __builtin_amdgcn_s_setprio(LDPC_AMD_DECODE_PRIO);
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;

    if (tid == 0)
        misc[0] = 0;

    // The work list of this wave with the VN block descriptors in place (plan.cpp, vn_work_desc): one scalar load per
    // block, none dependent on another (block id -> descriptor costs a second, dependent one).
    const auto my_vdesc = uniform_table(P.vn_work_desc + wave * (P.vn_work_stride + 1) * 4);
    auto vn_desc = [&](int w) { // count 0 = none (every row ends in one)
        const uint32_t d0 = my_vdesc[4 * w], d1 = my_vdesc[4 * w + 1], d2 = my_vdesc[4 * w + 2];
        return VnBlock{d0, d1, static_cast<uint16_t>(d2 & 0xFFFFu), static_cast<uint16_t>(d2 >> 16)};
    };
    // (kLlrRegs; RATIO, MINSUM) the slot indices a lane keeps in registers — two u16 per word for its nodes of degree <= 2,
    // 16 x u16 in eight words for its node in the wave's first block — are a property of the code: their loads go out
    // BEFORE the channel's, so that the two round trips to memory overlap (they used to follow each other, block by block)
    uint32_t my_idx[VNB + 1];
    uint32_t wide_idx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        wide_idx[i] = 0;
#pragma unroll
    for (int w = 0; w <= VNB; ++w)
        my_idx[w] = 0;
    // BS ("byte slots"): the instantiations compiled for small codes (the launcher: 8 * nnz < 2^16) keep byte offsets in
    // the 16-bit fields instead of element indices (slot_ptr)
    constexpr bool BS = LLR_MODE == kLlrRegs && (RATIO || MINSUM) && VNB != kMaxVnBlocksInRegs;
    constexpr int kSlotShift = BS ? 3 : 0;
    constexpr int kWideMax = BS ? 15 : 16; // widest node whose slot indices the lane keeps in registers (vn_block_ratio_regs)
    auto pick_up_indices = [&] {
        // as the lane keeps them, from the plan's table (plan.hpp, vn_packed): fifteen loads, none waiting for another (read
        // off the slot table they were sixteen dependent round trips for the wave that holds the wide block: 7 000 cycles by
        // which that wave reached the first barrier after the others)
        const uint32_t *pk = P.vn_packed + (static_cast<uint32_t>(wave) * kVnPackedRows) * kWaveSize + lane;
#pragma unroll
        for (int w = 0; w < VNB; ++w)
            my_idx[w] = pk[w * kWaveSize] << kSlotShift;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            wide_idx[i] = pk[(8 + i) * kWaveSize] << kSlotShift;
    };
    // (the hand-over instantiation picks them up after the channel: fifteen more live registers across the channel code
    // and its allocation falls apart, 36 -> 200 bytes of scratch)
    constexpr bool kIndicesInRegs = LLR_MODE == kLlrRegs && (RATIO || MINSUM);
    // kLaneChannel: every lane computes the input LLRs of its own variable nodes straight from the channel's data
    // (device_channel.hpp, channel_lanes): nothing is staged in LDS, no barrier stands between the channel and the first
    // variable-node work, and a frame's prologue is one round trip to the normals instead of a chain of them
    constexpr bool kLaneChannel = kIndicesInRegs && !HANDOVER;
    double my_llr[VNB + 1]; // (my_llr, my_idx: one constant entry of padding — the lock-step loops below name block w + 1, which an odd VNB does not have)
    my_llr[VNB] = 0.0;

    // ---- channel + LLR initialisation (device_channel.hpp) ----
    if constexpr (kLaneChannel)
    {
        const uint32_t *lane_pk = P.vn_packed + (static_cast<uint32_t>(wave) * kVnPackedRows) * kWaveSize + lane;
        channel_lanes<VNB>(a, frame, lane_pk, my_llr);
        // the slot indices now: their loads fly while the LLRs below become lambda (issued before the channel's loads they
        // would sit in fifteen registers across them, which the kernels at 96 do not have)
        pick_up_indices();
        if (a.llr_in_dump)
        {
            double *o = a.llr_in_dump + frame * nc;
#pragma unroll
            for (int w = 0; w < VNB; ++w)
                if (const uint32_t col = lane_pk[(24 + w) * kWaveSize]; col != kVnSrcZero)
                    o[col] = my_llr[w];
        }
    }
    else
        channel_init<kThreads>(a, frame, llr, tid);
#ifdef LDPC_AMD_PHASE_TRACE
    const uint64_t tr_chan0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (!kLaneChannel)
        __syncthreads();
#ifdef LDPC_AMD_PHASE_TRACE
    const uint64_t tr_chan = __builtin_amdgcn_s_memtime();
#endif

    if constexpr (!kLaneChannel)
        if (a.llr_in_dump)
        {
            double *o = a.llr_in_dump + frame * nc;
            for (int r = tid; r < nc; r += kThreads)
                o[P.rank_col[r]] = llr[r];
        }

    uint32_t escaped = 0; // RATIO: running maximum of dm_ratio_key over the frame's checked values (detmath.h)
    [[maybe_unused]] int32_t ho_key = 0; // HANDOVER: running maximum of dm_handover_key over the variable-node totals
    if constexpr (RATIO && LLR_MODE != kLlrRegs)
    {
        // input LLRs become lambda = e^-L in place (isolated variable nodes keep their LLR: nothing multiplies it)
        for (int r = tid; r < nc; r += kThreads)
            if (P.rank_slot0[r] != kNoSlot)
            {
                const double L = llr[r];
                if (!(__builtin_fabs(L) <= DM_RATIO_LLR_LIMIT))
                    escaped = ~0u;
                llr[r] = dm_exp_clamped(0.0 - L);
            }
        __syncthreads();
    }
    if constexpr (kIndicesInRegs && HANDOVER)
        pick_up_indices();
    if constexpr (LLR_MODE == kLlrRegs)
    {
        // every block's LLR first, then every block's arithmetic, branch-free: seven exponentials in lock step instead of
        // seven load -> exponential -> division chains one after the other (a frame holds its place on the CU for as long
        // as this takes).  A lane without a node in block w computes on a zero that nothing reads.
        bool leaf[VNB];
#pragma unroll
        for (int w = 0; w < VNB; ++w)
        {
            leaf[w] = false;
            if constexpr (!kLaneChannel)
                my_llr[w] = 0.0;
            if (w < P.vn_work_stride)
            {
                const VnBlock b = vn_desc(w);
                leaf[w] = b.degree == 1; // wave-uniform
                if constexpr (!kLaneChannel)
                    if (lane < b.count)
                        my_llr[w] = llr[b.first + lane];
            }
        }
        if constexpr (RATIO)
        {
#pragma unroll
            for (int w = 0; w < VNB; ++w)
            {
                if (!(__builtin_fabs(my_llr[w]) <= DM_RATIO_LLR_LIMIT))
                    escaped = ~0u;
                my_llr[w] = dm_exp_clamped(0.0 - my_llr[w]);
            }
#pragma unroll
            for (int w = 0; w < VNB; ++w)
            {
                const double rho = dm_ratio_div(1.0, my_llr[w]); // a leaf keeps its channel ratio rho_ch instead (vn_leaf_ratio)
                my_llr[w] = leaf[w] ? rho : my_llr[w];
            }
        }
        if constexpr (!kLaneChannel)
            __syncthreads(); // every lane holds its LLRs: the message array may now be written
    }
    // (the check-node work list likewise: cn_work_desc)
    const auto my_cdesc = uniform_table(reinterpret_cast<const uint32_t *>(P.cn_work_desc + wave * P.cn_desc_stride));
    auto cn_desc = [&](int w) { // count 0 = none (every row ends in two)
        const uint32_t d0 = my_cdesc[2 * w], d1 = my_cdesc[2 * w + 1];
        return CnBlock{d0, static_cast<uint16_t>(d1 & 0xFFFFu), static_cast<uint16_t>(d1 >> 16)};
    };
    // the w-th VN block of this wave: body(block, input LLR of this lane's node)
    auto for_my_vn_blocks = [&](auto &&body) {
        if constexpr (LLR_MODE == kLlrRegs)
        {
#pragma unroll
            for (int w = 0; w < VNB; ++w)
            {
                if (w >= P.vn_work_stride)
                    break;
                const VnBlock b = vn_desc(w);
                if (b.count == 0)
                    break;
                if (lane < b.count)
                    body(b, my_llr[w]);
            }
        }
        else
        {
            for (int w = 0; w < P.vn_work_stride; ++w)
            {
                const VnBlock b = vn_desc(w);
                if (b.count == 0)
                    break;
                if (lane < b.count)
                    body(b, llr[b.first + lane]);
            }
        }
    };

    // ---- v2c initialisation: decoder.cpp:16-19 (a handed-over frame: its c2v messages of iteration resume_at) ----
    if (resuming)
    {
        const double *src = a.ws_handover + static_cast<uint64_t>(blockIdx.x) * nnz;
        for (int e = tid; e < nnz; e += kThreads)
            msg[e] = a.handover_llr ? src[e] : 0.0 - dm_log(__builtin_fabs(src[e])); // lambda -> LLR: one logarithm per message
                                                                                        // (the fused form hands over LLRs)
    }
    else
    {
        // RATIO: L is lambda(L_ch) (kLlrRegs, a leaf: already rho(L_ch)), the first v2c is rho(L_ch)
        auto first_v2c = [&](const VnBlock &b, double L) {
            return RATIO ? ((LLR_MODE == kLlrRegs && b.degree == 1) ? L : dm_ratio_div(1.0, L)) : L;
        };
        auto from_table = [&](const VnBlock &b, double L) {
            const uint32_t *idx = P.vn_slot + b.idx_off + lane;
            const double v0 = first_v2c(b, L);
            for (int p = 0; p < b.degree; ++p)
                msg[idx[p * b.count]] = v0;
        };
        if constexpr (kIndicesInRegs)
        {
            // through the slot indices the lane holds already (a rolled loop over the slot table: up to 16 dependent
            // round trips to memory in front of the first iteration)
#pragma unroll
            for (int w = 0; w < VNB; ++w)
            {
                if (w >= P.vn_work_stride)
                    break;
                const VnBlock b = vn_desc(w);
                if (b.count == 0)
                    break;
                if (lane >= b.count)
                    continue;
                if (b.degree >= 1 && b.degree <= 2)
                {
                    const double v0 = first_v2c(b, my_llr[w]);
                    *slot_ptr<BS>(msg, my_idx[w] & 0xFFFFu) = v0;
                    *slot_ptr<BS>(msg, my_idx[w] >> 16) = v0; // (degree 1: the same slot again)
                }
                else if (w == 0 && b.degree >= 3 && b.degree <= kWideMax)
                {
                    const double v0 = first_v2c(b, my_llr[w]);
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (q < b.degree)
                            *slot_ptr<BS>(msg, (wide_idx[q >> 1] >> (16 * (q & 1))) & 0xFFFFu) = v0;
                }
                else
                    from_table(b, my_llr[w]);
            }
        }
        else
            for_my_vn_blocks(from_table);
    }
    __syncthreads();

    double *out_llr = WANT_LLR ? a.llr_out + frame * nc : nullptr;
    uint32_t I = 0;
    if constexpr (RATIO)
    {
        // Likelihood-ratio form.  Loop I: the CN pass of iteration I, which also reads — from the sign bits of the
        // v2c messages — the syndrome of the hard decisions made by VN pass I-1; one barrier with the vote; then
        // VN pass I.  A frame that converged after VN pass I-1 (or ran out of iterations) has made one CN pass
        // too many, which nothing reads: two barriers per iteration instead of three, no hard-bit array.
        PHASE_TIMERS
#ifdef LDPC_AMD_PHASE_TRACE
        const uint64_t tr_loop0 = __builtin_amdgcn_s_memtime();
#endif
        for (;;)
        {
            PHASE_START
            uint32_t bad = 0;
            // LDS-resident decoder with early termination: shared-reciprocal check nodes (detmath.h; the oracle applies the
            // same rule, a property of the code)
            // (not in the hand-over kernel: its frames iterate on after they have converged, the denominator products
            // overflow before the hand-over threshold is reached, and every overflow is a frame decoded again — measured:
            // 31.7 ms per batch instead of 13.2)
            // (memory-resident decoder with early termination: degree-6 nodes share reciprocals, mode 2; their range check counts
            // only when the frame goes on to the variable-node pass, so it is kept apart until the checks below have passed)
            constexpr int SH = (HANDOVER || SEPARATE) ? 0 : (LDS_RESIDENT ? 1 : (MAXD >= 6 ? 2 : 0));
            [[maybe_unused]] uint32_t esc6 = 0;
            uint32_t &cn_esc = SH == 2 ? esc6 : escaped;
            // blocks two at a time where they match (plan.cpp deals each wave's blocks in degree order); both
            // descriptors arrive with one scalar load (plan.cpp, cn_work_desc)
            for (int w = 0; w < P.cn_work_stride; w += 2)
            {
                const uint32_t d0 = my_cdesc[2 * w], d1 = my_cdesc[2 * w + 1], d2 = my_cdesc[2 * w + 2], d3 = my_cdesc[2 * w + 3];
                const CnBlock b0{d0, static_cast<uint16_t>(d1 & 0xFFFFu), static_cast<uint16_t>(d1 >> 16)};
                const CnBlock b1{d2, static_cast<uint16_t>(d3 & 0xFFFFu), static_cast<uint16_t>(d3 >> 16)};
                if (b0.count == 0)
                    break;
                if (b1.count == 0)
                {
                    bad |= cn_block_ratio<MAXD, SH>(msg, b0, lane, cn_esc);
                    break;
                }
                if (b0.count == kWaveSize && b1.count == kWaveSize)
                    bad |= cn_pair_ratio<MAXD, SH>(msg, b0.off, b1.off, b0.degree, b1.degree, lane, cn_esc);
                else
                    bad |= cn_block_ratio<MAXD, SH>(msg, b0, lane, cn_esc) | cn_block_ratio<MAXD, SH>(msg, b1, lane, cn_esc);
            }
            const int ph = I & 1;
            int wave_vote = (__ballot(bad != 0) != 0) | ((__ballot(DM_RATIO_ESCAPED(escaped)) != 0) << 1);
            if constexpr (HANDOVER)
                wave_vote |= (__ballot(DM_HANDOVER_DUE(ho_key)) != 0) << 2;
            if (lane == 0)
                votes[ph][wave] = wave_vote;
            PHASE_TICK(tr_cn)
            __syncthreads();
            PHASE_TICK(tr_w1)
            int any = 0;
#pragma unroll
            for (int w = 0; w < kDecodeWaves; ++w)
                any |= votes[ph][w];
#ifdef LDPC_AMD_PHASE_TRACE
            if (((I > 0 && !(any & 1)) || I == a.iterations) && a.phase_trace && frame >= 30000 && frame < 32048 && lane == 0)
            {
                uint64_t *o = a.phase_trace + ((frame - 30000) * 4 + wave) * 8;
                o[0] = tr_cn, o[1] = tr_w1, o[2] = tr_vn, o[3] = tr_w2;
                o[4] = tr_loop0 - tr_entry, o[5] = __builtin_amdgcn_s_memtime() - tr_loop0, o[6] = ((tr_chan0 - tr_entry) << 32) | (tr_chan - tr_chan0);
            }
#endif
            if (any & 2) // checked before the syndrome: an escaped frame's hard decisions mean nothing
            {
                if constexpr (CHAIN)
                    return true; // (uniform: every thread has read the same votes)
                if (tid == 0)
                {
                    const uint32_t pos = atomicAdd(a.redo_count, 1u);
                    a.redo_list[pos] = static_cast<uint32_t>(frame);
                    if constexpr (HANDOVER)
                        a.redo_iter[pos] = 0xFFFFFFFFu; // from scratch
                }
                return false;
            }
            if (I > 0 && a.early_term && !(any & 1)) // decoder.cpp:66-72 after VN pass I-1
            {
                --I;
                break;
            }
            if (I == a.iterations)
                break;
            if constexpr (HANDOVER)
                if (any & 4) // a total of VN pass I-1 left the inner box: the LLR-domain form continues with VN pass I
                {
                    if (tid == 0)
                    {
                        const uint32_t pos = atomicAdd(a.redo_count, 1u);
                        a.redo_list[pos] = static_cast<uint32_t>(frame);
                        a.redo_iter[pos] = I;
                        misc[1] = static_cast<int>(pos);
                    }
                    __syncthreads();
                    double *dst = a.ws_handover + static_cast<uint64_t>(static_cast<uint32_t>(misc[1])) * nnz;
                    for (int e = tid; e < nnz; e += kThreads)
                        dst[e] = msg[e]; // c2v of iteration I as lambda (sign bit: a decision); the resuming kernel takes the logarithm
                    return false;
                }
            if constexpr (SH == 2)
                escaped = escaped > esc6 ? escaped : esc6; // voted on after the next check-node pass, ahead of its syndrome
            // ---- VN pass, APP and hard decision: decoder.cpp:48-64 ----
            if constexpr (LLR_MODE == kLlrRegs)
            {
                auto put_llr = [&](const VnBlock &b, double prod) {
                    if constexpr (HANDOVER)
                        if (b.degree > 1) // (a leaf's total feeds nothing back)
                    {
                        const int32_t k = dm_handover_key(prod);
                        ho_key = k > ho_key ? k : ho_key;
                    }
                    if constexpr (WANT_LLR)
                        out_llr[P.rank_col[b.first + lane]] = 0.0 - dm_log(prod);
                };
                // one block: degree 1 or 2 from its register-held indices, the wave's first block likewise when it
                // has up to 16 edges per node, anything else through the slot table
                auto one = [&](const VnBlock &b, int w, uint32_t sl, double lam) {
                    if (lane >= b.count || b.degree == 0)
                        return;
                    double prod;
                    if (b.degree == 1)
                        prod = vn_small_ratio<1, BS>(msg, sl, lam, escaped);
                    else if (b.degree == 2)
                        prod = vn_small_ratio<2, BS>(msg, sl, lam, escaped);
                    else if (w == 0 && b.degree <= kWideMax)
                        prod = vn_block_ratio_regs<HANDOVER, BS>(msg, wide_idx, b.degree, lam, escaped);
                    else
                        prod = vn_block_ratio(msg, P.vn_slot + b.idx_off + lane, b.count, b.degree, lam, escaped);
                    put_llr(b, prod);
                };
                // this wave's block descriptors come straight from its work list (plan.cpp, vn_work_desc): one scalar
                // load each, none of them dependent on another
                auto vdesc = [&](int w) {
                    const uint32_t d0 = my_vdesc[4 * w], d1 = my_vdesc[4 * w + 1], d2 = my_vdesc[4 * w + 2];
                    return VnBlock{d0, d1, static_cast<uint16_t>(d2 & 0xFFFFu), static_cast<uint16_t>(d2 >> 16)};
                };
#pragma unroll
                for (int w = 0; w < VNB; w += 2) // full low-degree blocks two at a time in lock step
                {                                               // (plan.cpp deals each wave's blocks in degree order)
                    if (w >= P.vn_work_stride)
                        break;
                    const VnBlock b0 = vdesc(w);
                    if (b0.count == 0)
                        break;
                    const VnBlock b1 = vdesc(w + 1 < P.vn_work_stride ? w + 1 : P.vn_work_stride); // (row ends in a "none")
                    if (b1.count == 0)
                    {
                        one(b0, w, my_idx[w], my_llr[w]);
                        break;
                    }
                    if (b0.degree == b1.degree && b0.degree >= 1 && b0.degree <= 2 && b0.count == kWaveSize &&
                        b1.count == kWaveSize)
                    {
                        double pa, pb;
                        if (b0.degree == 1)
                            vn_small_ratio2<1, BS>(msg, my_idx[w], my_idx[w + 1], my_llr[w], my_llr[w + 1], escaped, pa, pb);
                        else
                            vn_small_ratio2<2, BS>(msg, my_idx[w], my_idx[w + 1], my_llr[w], my_llr[w + 1], escaped, pa, pb);
                        put_llr(b0, pa);
                        put_llr(b1, pb);
                    }
                    else
                    {
                        one(b0, w, my_idx[w], my_llr[w]);
                        one(b1, w + 1, my_idx[w + 1], my_llr[w + 1]);
                    }
                }
            }
            else
                for_my_vn_blocks([&](const VnBlock &b, double lam) {
                    if (b.degree == 0)
                        return;
                    const double prod = vn_block_ratio(msg, P.vn_slot + b.idx_off + lane, b.count, b.degree, lam, escaped);
                    if constexpr (HANDOVER)
                        if (b.degree > 1) // (a leaf's total feeds nothing back)
                    {
                        const int32_t k = dm_handover_key(prod);
                        ho_key = k > ho_key ? k : ho_key;
                    }
                    if constexpr (WANT_LLR)
                        out_llr[P.rank_col[b.first + lane]] = 0.0 - dm_log(prod);
                });
            PHASE_TICK(tr_vn)
            __syncthreads();
            PHASE_TICK(tr_w2)
            ++I;
        }
    }
    else
    {
        if (resuming)
            I = resume_at;
        while (I < a.iterations)
        {
            // ---- CN pass: decoder.cpp:25-45 (a handed-over frame arrives with the c2v messages of its first pass) ----
            if (resuming && I == resume_at)
            {
            }
            else if constexpr (MINSUM) // latency-bound: full blocks two at a time (one LDS round trip for both)
            {
                for (int w = 0; w < P.cn_work_stride; w += 2)
                {
                    const CnBlock b0 = cn_desc(w), b1 = cn_desc(w + 1);
                    if (b0.count == 0)
                        break;
                    if (b1.count == kWaveSize && b0.count == kWaveSize &&
                        cn_pair<MINSUM, MAXD>(msg, b0.off, b1.off, b0.degree, b1.degree, lane))
                        continue;
                    cn_block<MINSUM, MAXD>(msg, b0, lane, scratch, scratch2);
                    if (b1.count == 0)
                        break;
                    cn_block<MINSUM, MAXD>(msg, b1, lane, scratch, scratch2);
                }
            }
            else
                for (int w = 0; w < P.cn_work_stride; ++w)
                {
                    const CnBlock b = cn_desc(w);
                    if (b.count == 0)
                        break;
                    cn_block<MINSUM, MAXD>(msg, b, lane, scratch, scratch2);
                }
            __syncthreads();

            // ---- VN pass, APP and hard decision: decoder.cpp:48-64 ----
            // the per-edge hard bits feed the syndrome check and the outputs: without early termination only the
            // last pass has to store them (one LDS byte store per edge and iteration less)
            // (min-sum instantiations only: the sum-product instantiation's register allocation tips over with the flag,
            // 128 -> 141 VGPRs, and it is arithmetic-bound anyway)
            const bool store_hb = !MINSUM || a.early_term || I + 1 == a.iterations;
            auto vn_one = [&](const VnBlock &b, double L) {
                if (lane >= b.count)
                    return;
                double out;
                if constexpr (MINSUM && LLR_MODE == kLlrRegs) // (the register-budgeted instantiations: eight edges at a time)
                    out = vn_block_llr_lean(msg, hb, P.vn_slot + b.idx_off + lane, b.count, b.degree, L, store_hb);
                else
                    out = vn_block_llr(msg, hb, P.vn_slot + b.idx_off + lane, b.count, b.degree, L, store_hb);
                if constexpr (WANT_LLR)
                    out_llr[P.rank_col[b.first + lane]] = out;
            };
            // (min-sum, kLlrRegs) the passes whose decisions nobody reads — one block: degree 1 or 2 from its register-held
            // indices, the wave's first block likewise when it has up to 16 edges per node, anything else through the slot table
            [[maybe_unused]] auto vn_one_regs = [&](const VnBlock &b, int w, double L) {
                if (lane >= b.count)
                    return;
                double out;
                if (b.degree >= 1 && b.degree <= 2)
                    out = b.degree == 1 ? vn_small_llr<1, BS>(msg, my_idx[w], L) : vn_small_llr<2, BS>(msg, my_idx[w], L);
                else if (w == 0 && b.degree >= 3 && b.degree <= kWideMax)
                    out = vn_block_llr_regs<BS>(msg, wide_idx, b.degree, L);
                else
                    out = vn_block_llr_lean(msg, hb, P.vn_slot + b.idx_off + lane, b.count, b.degree, L, false);
                if constexpr (WANT_LLR)
                    out_llr[P.rank_col[b.first + lane]] = out;
            };
            if (MINSUM && LLR_MODE == kLlrRegs && !store_hb)
            {
#pragma unroll
                for (int w = 0; w < VNB; w += 2) // full low-degree blocks two at a time in lock step
                {
                    if (w >= P.vn_work_stride)
                        break;
                    const VnBlock b0 = vn_desc(w);
                    if (b0.count == 0)
                        break;
                    const VnBlock b1 = vn_desc(w + 1 < P.vn_work_stride ? w + 1 : P.vn_work_stride);
                    if (b1.count == 0)
                    {
                        vn_one_regs(b0, w, my_llr[w]);
                        break;
                    }
                    if (b0.degree == b1.degree && b0.degree >= 1 && b0.degree <= 2 && b0.count == kWaveSize &&
                        b1.count == kWaveSize)
                    {
                        double o0, o1;
                        if (b0.degree == 1)
                            vn_small_llr2<1, BS>(msg, my_idx[w], my_idx[w + 1], my_llr[w], my_llr[w + 1], o0, o1);
                        else
                            vn_small_llr2<2, BS>(msg, my_idx[w], my_idx[w + 1], my_llr[w], my_llr[w + 1], o0, o1);
                        if constexpr (WANT_LLR)
                        {
                            out_llr[P.rank_col[b0.first + lane]] = o0;
                            out_llr[P.rank_col[b1.first + lane]] = o1;
                        }
                    }
                    else
                    {
                        vn_one_regs(b0, w, my_llr[w]);
                        vn_one_regs(b1, w + 1, my_llr[w + 1]);
                    }
                }
            }
            else if constexpr (MINSUM && LLR_MODE == kLlrRegs)
            {
#pragma unroll
                for (int w = 0; w < VNB; w += 2) // the same with the hard decisions stored, indices from the slot table
                {
                    if (w >= P.vn_work_stride)
                        break;
                    const VnBlock b0 = vn_desc(w);
                    if (b0.count == 0)
                        break;
                    const VnBlock b1 = vn_desc(w + 1 < P.vn_work_stride ? w + 1 : P.vn_work_stride);
                    if (b1.count == 0)
                    {
                        vn_one(b0, my_llr[w]);
                        break;
                    }
                    if (b0.degree == b1.degree && b0.degree >= 1 && b0.degree <= 2 && b0.count == kWaveSize &&
                        b1.count == kWaveSize)
                    {
                        const uint32_t *i0 = P.vn_slot + b0.idx_off + lane, *i1 = P.vn_slot + b1.idx_off + lane;
                        double o0, o1;
                        if (b0.degree == 1)
                            vn_update_llr2<1>(true, msg, hb, i0, i1, my_llr[w], my_llr[w + 1], o0, o1);
                        else
                            vn_update_llr2<2>(true, msg, hb, i0, i1, my_llr[w], my_llr[w + 1], o0, o1);
                        if constexpr (WANT_LLR)
                        {
                            out_llr[P.rank_col[b0.first + lane]] = o0;
                            out_llr[P.rank_col[b1.first + lane]] = o1;
                        }
                    }
                    else
                    {
                        vn_one(b0, my_llr[w]);
                        vn_one(b1, my_llr[w + 1]);
                    }
                }
            }
            else
                for_my_vn_blocks([&](const VnBlock &b, double L) { vn_one(b, L); });
            __syncthreads();

            // ---- syndrome early termination: decoder.cpp:66-72, decoder.h:47-64 ----
            if (a.early_term)
            {
                int bad = 0;
                for (int w = 0; w < P.cn_work_stride; ++w)
                {
                    const CnBlock b = cn_desc(w);
                    if (b.count == 0)
                        break;
                    if (lane < b.count)
                    {
                        int par = 0;
                        for (int j = 0; j < b.degree; ++j)
                            par ^= hb[b.off + j * b.count + lane];
                        bad |= par;
                    }
                }
                // workgroup-wide OR with one barrier: every wave posts its vote in a slot of the iteration's parity
                // (slots alternate, so the next iteration's votes cannot overtake a slow reader)
                const int ph = I & 1;
                const int wave_vote = __ballot(bad != 0) != 0;
                if (lane == 0)
                    votes[ph][wave] = wave_vote;
                __syncthreads();
                int any = 0;
    #pragma unroll
                for (int w = 0; w < kDecodeWaves; ++w)
                    any |= votes[ph][w];
                if (!any)
                    break;
            }
            ++I;
        }
    }

    // ---- outputs: iteration count (decoder.cpp:74-77), hard decisions, bit errors (ldpcsim.cpp:184-188) ----
    if (tid == 0 && a.iters)
        a.iters[frame] = I;
    const bool ran = a.iterations > 0;
    auto hard_of_rank = [&](int r) -> int {
        if (!ran)
            return 0; // mCO is still zero-initialised when no iteration ran
        uint32_t s0 = P.rank_slot0[r];
        auto edge_bit = [&](uint32_t sl) -> int {
            if constexpr (RATIO)
                return hi_word(msg[sl]) >> 31; // the decision rides in the sign bit of the node's messages
            else
                return hb[sl];
        };
        if constexpr (LLR_MODE == kLlrRegs)
            return edge_bit(s0); // this mode is only used for codes without isolated variable nodes
        else
            return s0 != kNoSlot ? edge_bit(s0) : static_cast<int>(llr[r] <= 0);
    };
    if (a.hard)
    {
        uint8_t *h = a.hard + frame * nc;
        for (int r = tid; r < nc; r += kThreads)
            h[P.rank_col[r]] = static_cast<uint8_t>(hard_of_rank(r));
    }
    if constexpr (WANT_LLR)
    {
        if (!ran)
            for (int r = tid; r < nc; r += kThreads)
                out_llr[P.rank_col[r]] = 0.0;
        else if constexpr (LLR_MODE != kLlrRegs) // isolated variable nodes never pass through a VN block with edges
            for (int r = tid; r < nc; r += kThreads)
                if (P.rank_slot0[r] == kNoSlot)
                    out_llr[P.rank_col[r]] = llr[r];
    }
    if (a.bit_errors)
    {
        int err = 0;
        for (int i = tid; i < P.n_bitpos; i += kThreads) // the reference walks the whole bit_pos vector
        {
            int est = hard_of_rank(P.tx_rank[i]);
            int tx = cw ? static_cast<int>(cw[P.bit_pos[i]]) : 0;
            err += est != tx;
        }
        err = wave_sum(err);
        if (lane == 0 && err)
            atomicAdd(&misc[0], err);
        __syncthreads();
        if (tid == 0)
            a.bit_errors[frame] = static_cast<uint32_t>(misc[0]);
    }
#ifdef LDPC_AMD_PHASE_TRACE
    if constexpr (RATIO)
        if (a.phase_trace && frame >= 30000 && frame < 32048 && lane == 0)
            a.phase_trace[((frame - 30000) * 4 + wave) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
    return false;
}

template <bool MINSUM, bool WANT_LLR, bool LDS_RESIDENT, int MAXD, int LLR_MODE, bool RATIO, bool SEPARATE = false>
__global__ __launch_bounds__(kThreads) void decode_kernel(const DecodeArgs a)
{
    decode_body<MINSUM, WANT_LLR, LDS_RESIDENT, MAXD, LLR_MODE, RATIO, false, SEPARATE>(a, blockIdx.x);
}

// LDS-resident decoder, sum-product with early termination: the frames the first launch handed back (a.redo_list_in), each
// by the ratio form with separately divided outputs and — only if it leaves the box there too — by the LLR-domain form, in
// this workgroup: one launch where there were two, and a small grid that WALKS the list where every frame of the batch had a
// workgroup that looked whether it was meant (16 us per launch of 65 536 workgroups, for a list that holds a frame or none).
template <bool WANT_LLR, int MAXD, int LLR_MODE>
__global__ __launch_bounds__(kThreads) void decode_kernel_list(const DecodeArgs a)
{
    const uint32_t n = *uniform_table(a.redo_count_in);
    for (uint32_t slot = blockIdx.x; slot < n; slot += gridDim.x)
    {
        if (decode_body<false, WANT_LLR, true, MAXD, LLR_MODE, true, false, true, kMaxVnBlocksInRegs, true>(a, slot))
        {
            __syncthreads(); // the next form re-initialises LDS words this one may still be reading
            decode_body<false, WANT_LLR, true, MAXD, LLR_MODE, false, false, false, kMaxVnBlocksInRegs, true>(a, slot);
        }
        __syncthreads();
    }
}

#ifndef LDPC_AMD_HANDOVER_WAVES
#define LDPC_AMD_HANDOVER_WAVES 5
#endif
template <bool WANT_LLR, int MAXD, int LLR_MODE, int VNB = kMaxVnBlocksInRegs>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(LDPC_AMD_HANDOVER_WAVES, 5))) void decode_kernel_handover(const DecodeArgs a)
{
    decode_body<false, WANT_LLR, true, MAXD, LLR_MODE, true, true, false, VNB>(a, blockIdx.x);
}

// The same body compiled for five waves per SIMD (at most 96 VGPRs): the instantiations that sit at that boundary
// anyway (narrow LDS-resident codes without the LLR output) are pinned there, so that a change that costs one or two
// registers spills them instead of silently losing the fifth resident frame of every CU (-8 %).
constexpr int kW5VnBlocks = 7;
template <bool MINSUM, bool WANT_LLR, bool LDS_RESIDENT, int MAXD, int LLR_MODE, bool RATIO>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5, 5))) void decode_kernel_w5(const DecodeArgs a)
{
    decode_body<MINSUM, WANT_LLR, LDS_RESIDENT, MAXD, LLR_MODE, RATIO, false, false, kW5VnBlocks>(a, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// BEC: erasure decoder over the alphabet {0, 1, 'E'} (decoder.cpp:91-192), channel fused
// (channel.cpp:199-229).  All state is bytes in LDS: msg[nnz], sym[nc] (decoder input), lout[nc].
//
// The reference runs the forward/backward recursion with
//   cn_update(l, r) = 'E' if either is 'E' else l xor r                      (decoder.h:152-155)
//   vn_update(l, r, x) = x if either equals x else 'E'                       (decoder.h:145-148)
// Both recursions have closed forms over the node's other edges, used here (integer alphabet: the
// results are the same values, not approximations):
//   check node, edge j: 'E' if any other input is 'E', else the xor of the other inputs;
//   erased VN of degree >= 3, edge j: x if any other input equals x, else 'E';
//   erased VN of degree 2: the other input unchanged; degree 1: see deg1_compat.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t bec_state_bytes(int nnz, int nc)
{
    return static_cast<uint32_t>(((nnz + 15) / 16) * 16 + 2 * (((nc + 15) / 16) * 16));
}

__global__ __launch_bounds__(kThreads) void bec_kernel(const BecArgs a)
{
    extern __shared__ double lds[];
    __shared__ int misc[4];
    const DevPlan &P = a.plan;
    const int nnz = P.nnz, nc = P.nc, nct = P.nct;
    // state bytes: LDS, or device memory (a.ws: bec_state_bytes() per frame) for codes beyond 160 KB
    uint8_t *msg = a.ws ? a.ws + static_cast<uint64_t>(blockIdx.x) * bec_state_bytes(nnz, nc) : reinterpret_cast<uint8_t *>(lds);
    uint8_t *sym = msg + ((nnz + 15) / 16) * 16;
    uint8_t *lout = sym + ((nc + 15) / 16) * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t frame = blockIdx.x;
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;
    auto cw_of_rank = [&](int r) -> uint8_t { return cw ? cw[P.rank_col[r]] : 0; };

    if (tid == 0)
        misc[0] = 0;
    // ---- channel: channel.cpp:199-229 ----
    if (a.raw)
    {
        for (int r = tid; r < nc; r += kThreads)
        {
            uint8_t k = P.rank_kind[r];
            if (k == 1)
                sym[r] = kErasure;
            else if (k == 2)
            {
                // channel.cpp:222 indexes the transmitted-symbol vector by the COLUMN index
                uint32_t col = P.rank_col[r];
                sym[r] = (col < static_cast<uint32_t>(nct) && cw) ? cw[P.bit_pos[col]] : 0;
            }
            else if (k == 3)
                sym[r] = 0; // never written by the channel: the decoder's initial zero, a known 0 bit
        }
        const uint64_t *raw = a.raw + frame * static_cast<uint64_t>(nct);
        for (int i = tid; i < nct; i += kThreads)
        {
            bool erased = canonical(raw[i]) < a.eps;
            uint8_t xb = cw ? cw[P.bit_pos[i]] : 0;
            sym[P.tx_rank[i]] = erased ? kErasure : xb;
        }
    }
    else
    {
        const uint8_t *in = a.symbols + frame * nc;
        for (int r = tid; r < nc; r += kThreads)
            sym[r] = in[P.rank_col[r]];
    }
    __syncthreads();
    if (a.llr_in_dump)
    {
        double *o = a.llr_in_dump + frame * nc;
        for (int r = tid; r < nc; r += kThreads)
            o[P.rank_col[r]] = static_cast<double>(sym[r]);
    }
    for (int r = tid; r < nc; r += kThreads)
        lout[r] = 0; // mLLROut starts zeroed

    // work lists with the block descriptors in place (plan.cpp): one scalar load per block, no dependent second one
    const auto my_vdesc = uniform_table(P.vn_work_desc + wave * (P.vn_work_stride + 1) * 4);
    const auto my_cdesc = uniform_table(reinterpret_cast<const uint32_t *>(P.cn_work_desc + wave * P.cn_desc_stride));
    auto vn_desc = [&](int w) { // count 0 = none (every row ends in one)
        const uint32_t d0 = my_vdesc[4 * w], d1 = my_vdesc[4 * w + 1], d2 = my_vdesc[4 * w + 2];
        return VnBlock{d0, d1, static_cast<uint16_t>(d2 & 0xFFFFu), static_cast<uint16_t>(d2 >> 16)};
    };
    auto cn_desc = [&](int w) { // count 0 = none (every row ends in two)
        const uint32_t d0 = my_cdesc[2 * w], d1 = my_cdesc[2 * w + 1];
        return CnBlock{d0, static_cast<uint16_t>(d1 & 0xFFFFu), static_cast<uint16_t>(d1 >> 16)};
    };
    // v2c init: decoder.cpp:96-99
    for (int w = 0; w < P.vn_work_stride; ++w)
    {
        const VnBlock b = vn_desc(w);
        if (b.count == 0)
            break;
        if (lane < b.count)
        {
            uint8_t L = sym[b.first + lane];
            const uint32_t *idx = P.vn_slot + b.idx_off + lane;
            for (int p = 0; p < b.degree; ++p)
                msg[idx[p * b.count]] = L;
        }
    }
    __syncthreads();

    uint32_t I = 0;
    while (I < a.iterations)
    {
        // ---- CN update: decoder.cpp:105-123 ----
        for (int w = 0; w < P.cn_work_stride; ++w)
        {
            const CnBlock b = cn_desc(w);
            if (b.count == 0)
                break;
            if (lane < b.count)
            {
                uint8_t *m = msg + b.off + lane;
                int n_e = 0, x = 0;
                for (int j = 0; j < b.degree; ++j)
                {
                    uint8_t v = m[j * b.count];
                    if (v == kErasure)
                        ++n_e;
                    else
                        x ^= (v != 0);
                }
                for (int j = 0; j < b.degree; ++j)
                {
                    uint8_t v = m[j * b.count];
                    uint8_t o;
                    if (v == kErasure)
                        o = (n_e == 1) ? static_cast<uint8_t>(x) : kErasure;
                    else
                        o = (n_e == 0) ? static_cast<uint8_t>(x ^ (v != 0)) : kErasure;
                    m[j * b.count] = o;
                }
            }
        }
        __syncthreads();
        // ---- VN update: decoder.cpp:126-167 ----
        int any_e = 0;
        for (int w = 0; w < P.vn_work_stride; ++w)
        {
            const VnBlock b = vn_desc(w);
            if (b.count == 0)
                break;
            if (lane < b.count)
            {
                const int r = b.first + lane;
                const uint32_t *idx = P.vn_slot + b.idx_off + lane;
                const uint8_t x = cw_of_rank(r);
                const int vw = b.degree;
                if (sym[r] != kErasure)
                {
                    for (int p = 0; p < vw; ++p)
                        msg[idx[p * b.count]] = x;
                    lout[r] = x;
                }
                else if (vw == 0)
                {
                    // no edges: the reference would index an empty neighbour list; keep the erasure
                    lout[r] = kErasure;
                }
                else if (vw == 1)
                {
                    uint8_t c0 = msg[idx[0]];
                    msg[idx[0]] = a.deg1_compat ? 0 : kErasure;
                    lout[r] = c0;
                }
                else if (vw == 2)
                {
                    uint8_t c0 = msg[idx[0]], c1 = msg[idx[b.count]];
                    msg[idx[0]] = c1;
                    msg[idx[b.count]] = c0;
                    lout[r] = (c0 == x || c1 == x) ? x : kErasure;
                }
                else
                {
                    int hits = 0;
                    for (int p = 0; p < vw; ++p)
                        hits += msg[idx[p * b.count]] == x;
                    for (int p = 0; p < vw; ++p)
                    {
                        const uint32_t s = idx[p * b.count];
                        int own = msg[s] == x;
                        msg[s] = (hits - own) > 0 ? x : kErasure;
                    }
                    lout[r] = hits > 0 ? x : kErasure;
                }
                any_e |= lout[r] == kErasure;
            }
        }
        // early termination when no erasure is left (decoder.cpp:169-186); also the barrier of the pass
        if (a.early_term)
        {
            if (!__syncthreads_or(any_e))
                break;
        }
        else
            __syncthreads();
        ++I;
    }
    __syncthreads();

    if (tid == 0 && a.iters)
        a.iters[frame] = I;
    const bool ran = a.iterations > 0;
    // mCO: decoder.cpp:137,165 — the true bit, or 1 when the VN is still erased (-gf2 is always 1)
    auto hard_of_rank = [&](int r) -> int {
        if (!ran)
            return 0;
        return lout[r] == kErasure ? 1 : cw_of_rank(r);
    };
    if (a.hard)
    {
        uint8_t *h = a.hard + frame * nc;
        for (int r = tid; r < nc; r += kThreads)
            h[P.rank_col[r]] = static_cast<uint8_t>(hard_of_rank(r));
    }
    if (a.llr_out)
    {
        double *o = a.llr_out + frame * nc;
        for (int r = tid; r < nc; r += kThreads)
            o[P.rank_col[r]] = static_cast<double>(lout[r]);
    }
    if (a.bit_errors)
    {
        int err = 0;
        for (int i = tid; i < P.n_bitpos; i += kThreads)
        {
            int r = P.tx_rank[i];
            err += hard_of_rank(r) != static_cast<int>(cw_of_rank(r));
        }
        err = wave_sum(err);
        if (lane == 0 && err)
            atomicAdd(&misc[0], err);
        __syncthreads();
        if (tid == 0)
            a.bit_errors[frame] = static_cast<uint32_t>(misc[0]);
    }
}

// ---------------------------------------------------------------------------------------------
// encoder (see EncodeArgs)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void encode_info_kernel(const EncodeArgs a)
{
    // one thread per (frame, word): 64 bernoulli(0.5) draws -> one packed word
    const uint64_t gid = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (gid >= a.n_frames * static_cast<uint64_t>(a.words))
        return;
    const uint64_t f = gid / a.words;
    const int w = static_cast<int>(gid % a.words);
    const uint64_t *raw = a.info_raw + f * static_cast<uint64_t>(a.kc) + 64 * w;
    const int nb = min(64, a.kc - 64 * w);
    uint64_t bits = 0;
    for (int i = 0; i < nb; ++i)
        bits |= static_cast<uint64_t>(canonical(raw[i]) < 0.5) << i;
    a.prefix[gid] = bits;
}

// running XOR over frames, one workgroup per packed word column
__global__ __launch_bounds__(1024) void encode_prefix_kernel(const EncodeArgs a)
{
    __shared__ uint64_t part[1024];
    const int w = blockIdx.x, tid = threadIdx.x;
    const uint64_t per = (a.n_frames + 1023) / 1024;
    const uint64_t lo = min(tid * per, a.n_frames), hi = min(lo + per, a.n_frames);
    uint64_t s = 0;
    for (uint64_t f = lo; f < hi; ++f)
        s ^= a.prefix[f * a.words + w];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1)
    {
        uint64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] ^= v;
        __syncthreads();
    }
    uint64_t run = tid ? part[tid - 1] : 0;
    for (uint64_t f = lo; f < hi; ++f)
    {
        run ^= a.prefix[f * a.words + w];
        a.prefix[f * a.words + w] = run;
    }
}

// codeword[f][j] = cw_prev[j] ^ parity(prefix_f restricted to the rows of column j of G)
__global__ __launch_bounds__(256) void encode_cw_kernel(const EncodeArgs a, uint64_t first_frame)
{
    extern __shared__ uint64_t pw[];
    const uint64_t f = first_frame + blockIdx.x;
    for (int w = threadIdx.x; w < a.words; w += 256)
        pw[w] = a.prefix[f * a.words + w] ^ (a.base ? a.base[w] : 0ull);
    __syncthreads();
    const bool last = f + 1 == a.n_frames;
    uint8_t *out = a.codeword ? a.codeword + f * a.nc : nullptr;
    for (int j = threadIdx.x; j < a.nc; j += 256)
    {
        uint8_t b = a.cw_prev[j];
        if (j < a.g_cols)
            for (uint32_t p = a.g_col_ptr[j]; p < a.g_col_ptr[j + 1]; ++p)
            {
                uint32_t r = a.g_col_row[p];
                b ^= static_cast<uint8_t>(pw[r >> 6] >> (r & 63) & 1);
            }
        if (out)
            out[j] = b;
        if (last)
            a.cw_last[j] = b;
    }
}

// the same with the columns of G as bit masks (EncodeArgs::g_mask): a workgroup takes kEncFrames consecutive frames, a thread
// keeps the masks of its columns in registers (W words each) and the frames' prefixes arrive as scalars — a codeword bit is
// W ANDs and a population count instead of a walk over the column's entries with a bit test each (the walk: 3.5 ms per
// 65 536 frames of the n = 1024 code, more than the decode launch it feeds)
constexpr int kEncFrames = 32, kEncCols = 8; // columns per thread the kernel provides for: nc <= 256 * kEncCols
template <int W>
__global__ __launch_bounds__(256) void encode_cw_dense_kernel(const EncodeArgs a, uint64_t first_frame, uint64_t n_do)
{
    uint64_t m[kEncCols][W];
    uint8_t prev[kEncCols];
#pragma unroll
    for (int c = 0; c < kEncCols; ++c)
    {
        const int j = threadIdx.x + 256 * c;
        prev[c] = j < a.nc ? a.cw_prev[j] : 0;
#pragma unroll
        for (int w = 0; w < W; ++w)
            m[c][w] = j < a.nc ? a.g_mask[static_cast<size_t>(j) * W + w] : 0;
    }
    uint64_t base[W];
#pragma unroll
    for (int w = 0; w < W; ++w)
        base[w] = a.base ? uniform_table(a.base)[w] : 0ull;
    const uint64_t f0 = first_frame + static_cast<uint64_t>(blockIdx.x) * kEncFrames;
    const auto pre = uniform_table(a.prefix);
    for (int k = 0; k < kEncFrames; ++k)
    {
        const uint64_t f = f0 + k;
        if (f >= first_frame + n_do)
            break;
        uint64_t p[W];
#pragma unroll
        for (int w = 0; w < W; ++w)
            p[w] = pre[f * W + w] ^ base[w];
        const bool last = f + 1 == a.n_frames;
        uint8_t *out = a.codeword ? a.codeword + f * a.nc : nullptr;
#pragma unroll
        for (int c = 0; c < kEncCols; ++c)
        {
            const int j = threadIdx.x + 256 * c;
            if (j >= a.nc)
                break;
            uint64_t x = 0;
#pragma unroll
            for (int w = 0; w < W; ++w)
                x ^= p[w] & m[c][w];
            const uint8_t b = prev[c] ^ static_cast<uint8_t>(__popcll(x) & 1);
            if (out)
                out[j] = b;
            if (last)
                a.cw_last[j] = b;
        }
    }
}

// one workgroup sums the per-frame outputs of a batch (64 K frames: 64 per thread) into the five counters of the
// simulation loop (ldpcsim.cpp:175-200): frames, frame errors, bit errors, iterations, early stops
__global__ __launch_bounds__(1024) void batch_counters_kernel(const uint32_t *iters, const uint32_t *bit_errors, uint64_t n,
                                                              uint32_t max_iters, int early_term, long long *counters)
{
    __shared__ long long part[4][16];
    long long fe = 0, be = 0, it = 0, es = 0;
    auto take = [&](uint32_t b, uint32_t t) { fe += b > 0, be += b, it += t, es += early_term && t < max_iters; };
    // four frames per load, four loads in flight per array: the kernel sits between two batches' decode launches, and 64
    // dependent round trips per thread (one frame per load) were 35 us of every step
    uint64_t done = 0;
    if ((reinterpret_cast<uintptr_t>(iters) | reinterpret_cast<uintptr_t>(bit_errors)) % 16 == 0)
    {
        const uint4 *b4 = reinterpret_cast<const uint4 *>(bit_errors), *t4 = reinterpret_cast<const uint4 *>(iters);
        const uint64_t n4 = n / 4;
        uint64_t i = threadIdx.x;
        for (; i + 3 * 1024 < n4; i += 4 * 1024)
        {
            uint4 b[4], t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                b[k] = b4[i + k * 1024], t[k] = t4[i + k * 1024];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                take(b[k].x, t[k].x), take(b[k].y, t[k].y), take(b[k].z, t[k].z), take(b[k].w, t[k].w);
        }
        for (; i < n4; i += 1024)
        {
            const uint4 b = b4[i], t = t4[i];
            take(b.x, t.x), take(b.y, t.y), take(b.z, t.z), take(b.w, t.w);
        }
        done = n4 * 4;
    }
    for (uint64_t i = done + threadIdx.x; i < n; i += 1024)
        take(bit_errors[i], iters[i]);
    auto wave_total = [](long long v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            v += __shfl_xor(v, o, 64);
        return v;
    };
    fe = wave_total(fe), be = wave_total(be), it = wave_total(it), es = wave_total(es);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        part[0][wave] = fe, part[1][wave] = be, part[2][wave] = it, part[3][wave] = es;
    __syncthreads();
    if (threadIdx.x < 4)
    {
        long long s = 0;
        for (int w = 0; w < 16; ++w)
            s += part[threadIdx.x][w];
        counters[1 + threadIdx.x] = s;
    }
    if (threadIdx.x == 4)
        counters[0] = static_cast<long long>(n);
}

// operands a = 2^ea * ma, b = 2^eb * mb with ea, eb in [-500, 500] and random mantissas: a, b, a/b inside 2^-+1001
__global__ __launch_bounds__(256) void division_selftest_kernel(uint64_t n, uint64_t seed, unsigned long long *mismatches)
{
    unsigned long long bad = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
    {
        uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull ^ seed;
        auto next = [&] {
            x ^= x >> 12, x ^= x << 25, x ^= x >> 27;
            return x * 0x2545F4914F6CDD1Dull;
        };
        auto operand = [&] {
            const uint64_t m = next() >> 12, e = 1023 - 500 + (next() >> 33) % 1001;
            return dm_from_bits((e << 52) | m);
        };
        const double a = operand(), b = operand();
        volatile double bv = b; // keep the compiler from folding the two forms together
        bad += dm_bits(dm_ratio_div(a, b)) != dm_bits(a / bv);
        // dm_div_by (detmath.h): numerators of either sign up to 2^8 over divisors in [2^-7, 2^7] with THEIR correctly rounded
        // reciprocal (here: the device's own IEEE division, which is correctly rounded) — the channel's 2 y / sigma^2
        const uint64_t m2 = next() >> 12, e2 = 1023 - 7 + (next() >> 33) % 15;
        const double d = dm_from_bits((e2 << 52) | m2);
        volatile double dv = d;
        const double rcp = 1.0 / dv;
        const uint64_t m3 = next() >> 12, e3 = 1023 - 60 + (next() >> 33) % 69, s3 = next() >> 63;
        const double num = dm_from_bits((s3 << 63) | (e3 << 52) | m3);
        bad += dm_bits(dm_div_by(num, d, rcp)) != dm_bits(num / dv);
    }
    if (bad)
        atomicAdd(mismatches, bad);
}

template <bool LDS_RESIDENT, int MAXD, int LLR_MODE>
int launch_decode_impl(const DecodeArgs &a, bool min_sum, uint32_t lds_bytes, void *stream)
{
    const bool want_llr = a.llr_out != nullptr;
    // LDS-resident: the launch over the first launch's list (ratio_separate, a list coming in, none going out) is the chain
    // kernel decode_kernel_list; memory-resident: a separately dividing ratio launch with lists on both sides
    const bool list_chain = LDS_RESIDENT && a.ratio_separate && a.redo_count_in && !a.redo_list;
    if (list_chain && (min_sum || !a.early_term || a.iterations == 0 || !a.redo_list_in || a.redo_iter_in))
        return hipErrorInvalidValue;
    const bool ratio = a.redo_list != nullptr;
    // without early termination the ratio form runs with the hand-over to the LLR-domain form (detmath.h "Hand-over")
    const bool handover = ratio && !a.early_term;
    if (ratio && (min_sum || a.iterations == 0 || !a.redo_count || (a.redo_count_in && !a.ratio_separate) ||
                  (handover && (!a.redo_iter || !a.ws_handover))))
        return hipErrorInvalidValue;
    if (a.ratio_separate && !list_chain && (!ratio || handover || LDS_RESIDENT || MAXD < 6))
        return hipErrorInvalidValue;
    if (a.redo_iter_in && !a.ws_handover)
        return hipErrorInvalidValue;
    void (*k)(const DecodeArgs) = nullptr;
    if (min_sum)
        k = want_llr ? decode_kernel<true, true, LDS_RESIDENT, MAXD, LLR_MODE, false>
                     : decode_kernel<true, false, LDS_RESIDENT, MAXD, LLR_MODE, false>;
    else if (ratio)
    {
        k = want_llr ? decode_kernel<false, true, LDS_RESIDENT, MAXD, LLR_MODE, true>
                     : decode_kernel<false, false, LDS_RESIDENT, MAXD, LLR_MODE, true>;
    }
    else
    {
        k = want_llr ? decode_kernel<false, true, LDS_RESIDENT, MAXD, LLR_MODE, false>
                     : decode_kernel<false, false, LDS_RESIDENT, MAXD, LLR_MODE, false>;
    }
    // the instantiations for small codes: at most kW5VnBlocks VN blocks per wave, message byte offsets within 16 bits
    [[maybe_unused]] const bool few_vn_blocks = a.plan.vn_work_stride <= kW5VnBlocks && 8 * a.plan.nnz < 65536;
    if constexpr (LDS_RESIDENT && MAXD == 4 && LLR_MODE == kLlrRegs)
        if (!want_llr && few_vn_blocks)
        {
            if (min_sum)
                k = decode_kernel_w5<true, false, LDS_RESIDENT, MAXD, LLR_MODE, false>;
            else if (ratio)
                k = decode_kernel_w5<false, false, LDS_RESIDENT, MAXD, LLR_MODE, true>;
        }
    if constexpr (LDS_RESIDENT)
    {
        if (handover)
        {
            k = want_llr ? decode_kernel_handover<true, MAXD, LLR_MODE> : decode_kernel_handover<false, MAXD, LLR_MODE>;
            if constexpr (MAXD == 4 && LLR_MODE == kLlrRegs)
                if (few_vn_blocks)
                    k = want_llr ? decode_kernel_handover<true, MAXD, LLR_MODE, kW5VnBlocks> : decode_kernel_handover<false, MAXD, LLR_MODE, kW5VnBlocks>;
        }
        else if (list_chain)
            k = want_llr ? decode_kernel_list<true, MAXD, LLR_MODE> : decode_kernel_list<false, MAXD, LLR_MODE>;
    }
    if constexpr (!LDS_RESIDENT && MAXD >= 6)
        if (ratio && a.ratio_separate)
            k = want_llr ? decode_kernel<false, true, false, MAXD, LLR_MODE, true, true> : decode_kernel<false, false, false, MAXD, LLR_MODE, true, true>;
    if (handover && !LDS_RESIDENT)
        return hipErrorInvalidValue; // (the memory-resident decoder runs the LLR-domain form when early termination is off)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess)
        return e;
    const unsigned grid = list_chain ? static_cast<unsigned>(std::min<uint64_t>(a.n_frames, 1024)) : static_cast<unsigned>(a.n_frames);
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds_bytes, static_cast<hipStream_t>(stream), a);
    return hipGetLastError();
}

template <bool LDS_RESIDENT, int MAXD, int LLR_MODE>
int launch_decode(const DecodeArgs &a, bool min_sum, uint32_t lds_bytes, void *stream)
{
#ifdef LDPC_AMD_HEADLINE_SHAPE_ONLY // register-allocation experiments: one code shape, a fraction of the build time
    if constexpr (!(LDS_RESIDENT && MAXD == 4 && LLR_MODE == kLlrRegs))
        return hipErrorInvalidValue;
    else
#endif
        return launch_decode_impl<LDS_RESIDENT, MAXD, LLR_MODE>(a, min_sum, lds_bytes, stream);
}

} // namespace

int launch_decode_lds(const DecodeArgs &a, bool min_sum, int max_cn_degree, int llr_mode, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    if (llr_mode == kLlrMem && !a.ws_llr)
        return hipErrorInvalidValue;
    if (llr_mode == kLlrRegs && (a.plan.vn_work_stride > kMaxVnBlocksInRegs || a.plan.nc > a.plan.nnz || !a.plan.vn_packed))
        return hipErrorInvalidValue;
    uint32_t lds = a.plan.lds_bytes - (llr_mode != kLlrLds ? 8u * static_cast<uint32_t>(a.plan.nc) : 0u);
    if (a.redo_list) // ratio form: no hard-bit array (the last array of the frame in every layout)
        lds -= ((static_cast<uint32_t>(a.plan.nnz) + 15u) / 16u) * 16u;
#define LDPC_PICK(D)                                                              \
    switch (llr_mode)                                                             \
    {                                                                             \
    case kLlrMem: return launch_decode<true, D, kLlrMem>(a, min_sum, lds, stream);   \
    case kLlrRegs: return launch_decode<true, D, kLlrRegs>(a, min_sum, lds, stream); \
    default: return launch_decode<true, D, kLlrLds>(a, min_sum, lds, stream);        \
    }
    if (max_cn_degree <= 4)
    {
        LDPC_PICK(4)
    }
    if (max_cn_degree <= 8)
    {
        LDPC_PICK(8)
    }
#undef LDPC_PICK
    return hipErrorInvalidValue;
}

int launch_decode_mem(const DecodeArgs &a, bool min_sum, int max_cn_degree, uint32_t occupancy_lds, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    if (!a.ws_msg || !a.ws_llr || !a.ws_hb)
        return hipErrorInvalidValue;
    if (max_cn_degree <= 4)
        return launch_decode<false, 4, kLlrMem>(a, min_sum, occupancy_lds, stream);
    if (max_cn_degree <= 8)
        return launch_decode<false, 8, kLlrMem>(a, min_sum, occupancy_lds, stream);
    if (max_cn_degree > 16 && (!a.ws_scr || a.redo_list)) // wide nodes: scratch needed, no likelihood-ratio form
        return hipErrorInvalidValue;
    return launch_decode<false, 16, kLlrMem>(a, min_sum, occupancy_lds, stream);
}

int launch_batch_counters(const uint32_t *iters, const uint32_t *bit_errors, uint64_t n, uint32_t max_iters, int early_term,
                          long long *counters, void *stream)
{
    hipLaunchKernelGGL(batch_counters_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), iters, bit_errors, n,
                       max_iters, early_term, counters);
    return hipGetLastError();
}

int launch_division_selftest(uint64_t n, uint64_t seed, unsigned long long *mismatches, void *stream)
{
    if (n == 0)
        return hipSuccess;
    const unsigned blocks = static_cast<unsigned>(std::min<uint64_t>((n + 255) / 256, 8192));
    hipLaunchKernelGGL(division_selftest_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), n, seed,
                       mismatches);
    return hipGetLastError();
}

int launch_bec(const BecArgs &a, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    // codes whose bit-sliced state fits LDS: 64 frames per workgroup (kernels_bec.hip); LDPC_AMD_NO_BEC_SLICED: experiments
    if (!a.ws && bec_sliced_fits(a.plan) && !std::getenv("LDPC_AMD_NO_BEC_SLICED"))
        return launch_bec_sliced(a, stream);
    const uint32_t lds = a.ws ? 16u : bec_state_bytes(a.plan.nnz, a.plan.nc);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bec_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(bec_kernel, dim3(static_cast<unsigned>(a.n_frames)), dim3(kThreads), lds,
                       static_cast<hipStream_t>(stream), a);
    return hipGetLastError();
}

int launch_encode_prefix(const EncodeArgs &a, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t items = a.n_frames * static_cast<uint64_t>(a.words);
    hipLaunchKernelGGL(encode_info_kernel, dim3(static_cast<unsigned>((items + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(encode_prefix_kernel, dim3(a.words), dim3(1024), 0, s, a);
    return hipGetLastError();
}

int launch_encode_codewords(const EncodeArgs &a, void *stream, bool only_last)
{
    if (a.n_frames == 0)
        return hipSuccess;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = sizeof(uint64_t) * a.words;
    const bool all = !only_last && a.codeword;
    if (a.g_mask && a.words <= 4 && a.nc <= 256 * kEncCols)
    {
        const uint64_t first = all ? 0 : a.n_frames - 1, n_do = all ? a.n_frames : 1;
        const dim3 grid(static_cast<unsigned>((n_do + kEncFrames - 1) / kEncFrames));
        switch (a.words)
        {
        case 1: hipLaunchKernelGGL(encode_cw_dense_kernel<1>, grid, dim3(256), 0, s, a, first, n_do); break;
        case 2: hipLaunchKernelGGL(encode_cw_dense_kernel<2>, grid, dim3(256), 0, s, a, first, n_do); break;
        case 3: hipLaunchKernelGGL(encode_cw_dense_kernel<3>, grid, dim3(256), 0, s, a, first, n_do); break;
        default: hipLaunchKernelGGL(encode_cw_dense_kernel<4>, grid, dim3(256), 0, s, a, first, n_do); break;
        }
        return hipGetLastError();
    }
    if (all)
        hipLaunchKernelGGL(encode_cw_kernel, dim3(static_cast<unsigned>(a.n_frames)), dim3(256), lds, s, a, uint64_t(0));
    else
        hipLaunchKernelGGL(encode_cw_kernel, dim3(1), dim3(256), lds, s, a, a.n_frames - 1);
    return hipGetLastError();
}

int launch_encode(const EncodeArgs &a, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    int rc = launch_encode_prefix(a, stream);
    if (rc != hipSuccess)
        return rc;
    return launch_encode_codewords(a, stream, a.codeword == nullptr); // (no codewords wanted: only the running one after the batch)
}

} // namespace ldpc_amd
