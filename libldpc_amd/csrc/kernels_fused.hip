// kernels_fused.hip — the fused form of the likelihood-ratio iteration (detmath.h "Fused form", fused_rule.h, plan.hpp
// FusedPlan): first launch of sum-product with early termination for codes whose check nodes have 2..4 edges and at most
// one leaf each — the n = 1024 test code of the reference and the headline workload.  One workgroup (4 waves) per frame,
// messages in LDS, hand-written for gfx950 / wave64.
//
// Reference semantics restated (file:line in heat1q/libldpc): decode loop src/decoding/decoder.cpp:11-78, box-plus
// src/decoding/decoder.h:12-15, syndrome early termination decoder.h:47-64, channel + LLRs src/sim/channel.cpp:62-93 /
// 129-162, bit-error count src/sim/ldpcsim.cpp:184-188.
//
// What differs from the general LDS-resident kernel (kernels.hip, decode_body<RATIO>):
//   * prologue: the frame's channel values are evaluated ONCE, spread evenly over the 256 threads — transmitted bit i by
//     thread i mod 256: LLR, lambda = e^-L, rho = 1 / lambda — and staged in the (still unused) message area, from where
//     every lane picks up the values of the nodes it serves.  (The general kernel lets every lane evaluate the seven block
//     slots of its wave whether they hold a node or not, with 64-bit slab arithmetic per value: 260 k lane-instructions per
//     frame of the n = 1024 code, a fifth of the kernel.  Here the slab addressing is wave-uniform scalar work.)
//   * a leaf (degree-1 variable node) has no message slot and no variable-node visit: its check node's lane keeps its
//     channel ratio in a register and takes its hard decision by one multiplication and a comparison;
//   * a variable node of degree 2 multiplies (its check nodes hand it rho(c2v)): no division;
//   * hard decisions live in registers of the lanes that make them (one bit per block slot), so check-node outputs carry
//     no sign and the epilogue reads no message;
//   * the box check of the v2c messages is a running max3 / min3 over their upper words.
// Frames whose values leave the box (or whose denominator products leave theirs) go to a.redo_list, as in the general
// kernel; the second launch (separately divided ratio form, kernels.hip) decodes them from scratch.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_cn.hpp"
#include "device_math.hpp"
#include "fused_rule.h"
#include "kernels.hpp"

#ifndef LDPC_AMD_DECODE_PRIO
#define LDPC_AMD_DECODE_PRIO 3
#endif
#ifndef LDPC_AMD_FUSED_WAVES
#define LDPC_AMD_FUSED_WAVES 6
#endif

namespace ldpc_amd
{

namespace
{

constexpr int kThreads = kDecodeWaves * kWaveSize;

__device__ __forceinline__ uint32_t hi_word(double x) { return static_cast<uint32_t>(dm_bits(x) >> 32); }
__device__ __forceinline__ double with_sign(double mag, uint32_t sign_hi) // mag > 0, sign_hi = 0 or 0x80000000
{
    return dm_from_bits(dm_bits(mag) | (static_cast<uint64_t>(sign_hi) << 32));
}
__device__ __forceinline__ double *at(char *msg, uint32_t byte_off) { return reinterpret_cast<double *>(msg + byte_off); }

// running extremes of the upper words of the v2c messages (detmath.h, dm_box_escaped) and of the inverted products
struct Track
{
    uint32_t hmax, hmin, pmax;
};
__device__ __forceinline__ void track(Track &t, double a)
{
    t.hmax = max(t.hmax, hi_word(a));
    t.hmin = min(t.hmin, hi_word(a));
}
__device__ __forceinline__ void track2(Track &t, double a, double b)
{
    t.hmax = max(max(t.hmax, hi_word(a)), hi_word(b)); // v_max3_u32
    t.hmin = min(min(t.hmin, hi_word(a)), hi_word(b)); // v_min3_u32
}

// ---- check nodes ------------------------------------------------------------------------------------------------------
// One call = one or two blocks (TWO: both full, 64 nodes) of one class <D, LEAF, FLIP>: M = D - LEAF message inputs per node
// at p + k * stride, the leaf's channel ratio in rho*.  prev* (bit 31): the leaf's decision of the previous pass, which
// joins the parity of the node; lb*: its new decision.  Returns the parity words (bit 31 counts) of the nodes, or-ed.
template <int D, bool LEAF, unsigned FLIP, bool TWO, bool WANT_TOT, bool SHARED = true>
__device__ __forceinline__ uint32_t cnf_call(char *msg, uint32_t off0, uint32_t off1, uint32_t cnt0, int lane, double rho0, double rho1,
                                             uint32_t prev0, uint32_t prev1, uint32_t &lb0, uint32_t &lb1, Track &t, double &tot0,
                                             double &tot1)
{
    constexpr int M = D - (LEAF ? 1 : 0);
    const uint32_t stride = TWO ? kWaveSize * 8u : cnt0 * 8u;
    char *p0 = msg + off0 + lane * 8, *p1 = msg + off1 + lane * 8;
    double v0[D], v1[D];
    uint32_t par0 = prev0, par1 = prev1;
#pragma unroll
    for (int k = 0; k < M; ++k)
    {
        const double x = *at(p0, k * stride);
        par0 ^= hi_word(x);
        v0[k] = __builtin_fabs(x);
    }
    if constexpr (TWO)
    {
#pragma unroll
        for (int k = 0; k < M; ++k)
        {
            const double x = *at(p1, k * stride);
            par1 ^= hi_word(x);
            v1[k] = __builtin_fabs(x);
        }
    }
    if constexpr (LEAF)
        v0[M] = rho0, v1[M] = rho1;
    auto node = [&](double(&v)[D], uint32_t &lb, double &tot) {
        uint32_t h;
        if constexpr (D == 2)
            h = dm_cnf2(v, FLIP);
        else if constexpr (D == 3)
            h = dm_cnf3(v, FLIP, LEAF, SHARED, &lb, WANT_TOT ? &tot : nullptr);
        else
            h = dm_cnf4(v, FLIP, LEAF, SHARED, &lb, WANT_TOT ? &tot : nullptr);
        t.pmax = max(t.pmax, h);
    };
    node(v0, lb0, tot0);
    if constexpr (TWO)
        node(v1, lb1, tot1);
#pragma unroll
    for (int k = 0; k < M; ++k)
        *at(p0, k * stride) = v0[k];
    if constexpr (TWO)
    {
#pragma unroll
        for (int k = 0; k < M; ++k)
            *at(p1, k * stride) = v1[k];
        return par0 | par1;
    }
    return par0;
}

// class keys (fused_rule.h, dm_fused_class): the flipped outputs are the last NM of the node's M message inputs
constexpr unsigned cls_key(int d, int leaf, int nm) { return static_cast<unsigned>(d) | (static_cast<unsigned>(leaf) << 3) | ((((1u << nm) - 1u) << (d - leaf - nm)) << 4); }
constexpr unsigned cls_flip(int d, int leaf, int nm) { return ((1u << nm) - 1u) << (d - leaf - nm); }

// ---- variable nodes ---------------------------------------------------------------------------------------------------
// degree 2, two nodes of two full blocks in lock step (detmath.h "Fused form", item 3): inputs rho(c2v), outputs
// rho(v2c_0) = rho_ch rho(c2v_1), rho(v2c_1) = rho_ch rho(c2v_0); rho(total) = rho(v2c_0) rho(c2v_0) <= 1: decision 1.
// Returns the sign words (bit 31 = the decision); ta / tb = rho(total).
__device__ __forceinline__ void vn2_pair(char *msg, uint32_t packed_a, uint32_t packed_b, double ra, double rb, Track &t, uint32_t &sga,
                                         uint32_t &sgb, double &ta, double &tb)
{
    asm volatile("" : "+v"(packed_a), "+v"(packed_b)); // unpack here, every pass (unpacked offsets would stay live)
    const uint32_t a0 = packed_a & 0xFFFFu, a1 = packed_a >> 16, b0 = packed_b & 0xFFFFu, b1 = packed_b >> 16;
    const double ca0 = __builtin_fabs(*at(msg, a0)), ca1 = __builtin_fabs(*at(msg, a1));
    const double cb0 = __builtin_fabs(*at(msg, b0)), cb1 = __builtin_fabs(*at(msg, b1));
    const double oa0 = ra * ca1, oa1 = ra * ca0, ob0 = rb * cb1, ob1 = rb * cb0;
    ta = oa0 * ca0, tb = ob0 * cb0;
    sga = ta <= 1.0 ? 0x80000000u : 0u, sgb = tb <= 1.0 ? 0x80000000u : 0u;
    track2(t, oa0, oa1);
    track2(t, ob0, ob1);
    *at(msg, a0) = with_sign(oa0, sga), *at(msg, a1) = with_sign(oa1, sga);
    *at(msg, b0) = with_sign(ob0, sgb), *at(msg, b1) = with_sign(ob1, sgb);
}

__device__ __forceinline__ void vn2_one(char *msg, uint32_t packed_a, double ra, Track &t, uint32_t &sga, double &ta)
{
    asm volatile("" : "+v"(packed_a));
    const uint32_t a0 = packed_a & 0xFFFFu, a1 = packed_a >> 16;
    const double ca0 = __builtin_fabs(*at(msg, a0)), ca1 = __builtin_fabs(*at(msg, a1));
    const double oa0 = ra * ca1, oa1 = ra * ca0;
    ta = oa0 * ca0;
    sga = ta <= 1.0 ? 0x80000000u : 0u;
    track2(t, oa0, oa1);
    *at(msg, a0) = with_sign(oa0, sga), *at(msg, a1) = with_sign(oa1, sga);
}

// degree DV in 3..15, slot offsets in registers (16 bits each, two per word): the wave's first block.  lambda(total) =
// lambda_ch prod lambda(c2v_p) in column file order (decoder.cpp:50-56), range-checked at every third factor when the
// product is longer than four; rho(v2c_p) = lambda(c2v_p) / lambda(total).  Returns lambda(total).
template <int DV, bool TWICE>
__device__ __forceinline__ double vn_wide(char *msg, const uint32_t (&packed)[8], double lam, Track &t, uint32_t &sg)
{
    uint32_t pk[(DV + 1) / 2];
#pragma unroll
    for (int i = 0; i < (DV + 1) / 2; ++i)
    {
        pk[i] = packed[i];
        asm volatile("" : "+v"(pk[i])); // the words stay packed across passes
    }
    auto slot = [&](int p) { return at(msg, (p & 1) ? pk[p >> 1] >> 16 : pk[p >> 1] & 0xFFFFu); };
    double c[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c[p] = __builtin_fabs(*slot(p));
    if constexpr (TWICE) // the offsets are unpacked again for the writes: fifteen addresses held across the product and the
    {                    // division are fifteen registers the instantiation at six waves per SIMD does not have
#pragma unroll
        for (int i = 0; i < (DV + 1) / 2; ++i)
            asm volatile("" : "+v"(pk[i]));
    }
    double prod = lam;
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        prod *= c[p];
        if (DV > 3 && p % 3 == 2)
            track(t, prod);
    }
    sg = prod >= 1.0 ? 0x80000000u : 0u;
    // the decision rides in the sign of rho(total): every product below comes out with it (same magnitudes, no per-edge
    // or); the node's extremes are taken over the signed upper words — all carry the same top bit — and stripped once
    const double tot = with_sign(dm_ratio_div(1.0, prod), sg);
    uint32_t nmax = 0u, nmin = 0xFFFFFFFFu;
#pragma unroll
    for (int p = 0; p + 1 < DV; p += 2)
    {
        const double o0 = tot * c[p], o1 = tot * c[p + 1];
        nmax = max(max(nmax, hi_word(o0)), hi_word(o1));
        nmin = min(min(nmin, hi_word(o0)), hi_word(o1));
        *slot(p) = o0, *slot(p + 1) = o1;
    }
    if constexpr (DV % 2 == 1)
    {
        const double o = tot * c[DV - 1];
        nmax = max(nmax, hi_word(o)), nmin = min(nmin, hi_word(o));
        *slot(DV - 1) = o;
    }
    t.hmax = max(t.hmax, nmax & 0x7FFFFFFFu);
    t.hmin = min(t.hmin, nmin & 0x7FFFFFFFu);
    return prod;
}

// any degree >= 3 through the plan's table of slot offsets (rolled: every message is read twice; blocks that are neither
// the wave's first nor of degree 2 — none in the n = 1024 code)
__device__ __forceinline__ double vn_table(char *msg, const uint32_t *tbl, int lane, int count, int degree, double lam, Track &t, uint32_t &sg)
{
    // (tbl is wave-uniform, the lane joins as a 32-bit offset in every load: a per-lane pointer would be hoisted out of the
    // decode loop and held — or spilled — for the whole decode)
    double prod = lam;
    for (int p = 0; p < degree; ++p)
    {
        prod *= __builtin_fabs(*at(msg, tbl[p * count + lane]));
        if (degree > 3 && p % 3 == 2)
            track(t, prod);
    }
    sg = prod >= 1.0 ? 0x80000000u : 0u;
    const double tot = dm_ratio_div(1.0, prod);
    for (int p = 0; p < degree; ++p)
    {
        double *m = at(msg, tbl[p * count + lane]);
        const double o = tot * __builtin_fabs(*m);
        track(t, o);
        *m = with_sign(o, sg);
    }
    return prod;
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// ---- the frame's channel values, one stage entry {lambda, rho} per transmitted bit (AWGN / BSC) or per column (given LLRs) ----
// AWGN: normal g of the stream is element (g & 1) of accepted polar pair g >> 1; the pairs lie in one slab per generator
// chunk (kernels.hpp, DecodeArgs).  Element i of the frame (g = g0 + i) sits at word p0[i + (i >= t1 ? d1 : 0) + (i >= t2 ?
// d2 : 0)]: everything but the two comparisons is wave-uniform.
struct AwgnFrame
{
    const uint64_t *p0;
    uint32_t t1, t2, d1, d2;
    bool ok;
};
__device__ __forceinline__ AwgnFrame awgn_frame(const DecodeArgs &a, uint64_t frame, int nct)
{
    AwgnFrame f;
    const uint64_t g0 = a.normal_base + frame * static_cast<uint64_t>(nct);
    const uint64_t q_lo = g0 >> 1, q_hi = (g0 + nct - 1) >> 1;
    const auto cum = uniform_table(a.slab_cum);
    const uint64_t rel_lo = q_lo - a.pair_origin, rel_hi = q_hi - a.pair_origin;
    uint32_t j0 = static_cast<uint32_t>(static_cast<float>(rel_lo) * a.slab_pairs_inv);
    j0 = j0 < a.n_slabs ? j0 : a.n_slabs - 1;
    uint32_t jb = j0 > 0 ? j0 - 1 : 0;
    uint64_t c0 = cum[jb], c1 = cum[jb + 1], c2 = cum[jb + 2], c3 = cum[jb + 3];
    if (!(c0 <= rel_lo && rel_hi < c3)) // (a guess off by more than one slab: walk the table)
    {
        while (j0 > 0 && rel_lo < cum[j0])
            --j0;
        while (j0 + 1 < a.n_slabs && rel_lo >= cum[j0 + 1])
            ++j0;
        jb = j0;
        c0 = cum[jb], c1 = cum[jb + 1], c2 = cum[jb + 2], c3 = cum[jb + 3];
    }
    f.ok = c0 <= rel_lo && rel_hi < c3; // (a frame over more than three slabs: left to the general kernel)
    auto threshold = [&](uint64_t c) -> uint32_t { // first i with ((g0 + i) >> 1) - origin >= c
        if (c <= rel_lo)
            return 0u;
        const uint64_t dlt = c - rel_lo;
        return dlt > 0x3FFFFFFFull ? 0x7FFFFFFFu : static_cast<uint32_t>(2 * dlt - (g0 & 1));
    };
    f.t1 = threshold(c1), f.t2 = threshold(c2);
    f.p0 = a.pairs + (static_cast<uint64_t>(jb) * a.slab_words + g0 - 2 * (a.pair_origin + c0));
    f.d1 = static_cast<uint32_t>(a.slab_words - 2 * (c1 - c0));
    f.d2 = static_cast<uint32_t>(a.slab_words - 2 * (c2 - c1));
    return f;
}

// ---- prologue, part 1 of both forms: the frame's input LLRs, one per transmitted bit (AWGN / BSC: entry i = transmitted bit i,
// entry nct = the value of a shortened bit) or per column (given LLRs), spread evenly over the workgroup's threads;
// put(entry, L) stages what the form keeps of it.  Returns false when the frame's normals cannot be addressed here
// (awgn_frame).  channel.cpp:62-93 / 129-162, shared.cpp:50-55.
template <typename Put>
__device__ __forceinline__ bool stage_channel(const DecodeArgs &a, const DevFusedPlan &F, uint64_t frame, int tid, Put &&put)
{
    const DevPlan &P = a.plan;
    const int nc = P.nc, nct = P.nct;
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;
    const bool given = a.mode == kModeLlr;
    const int n_stage = given ? nc : nct;
    bool ok = true;
    double *dump = a.llr_in_dump ? a.llr_in_dump + frame * nc : nullptr;
    if (dump && !given) // columns the channel does not write: punctured 0.0, shortened (channel.cpp:73-86)
        for (int r = tid; r < nc; r += kThreads)
            if (const uint8_t k = P.rank_kind[r]; k != 0)
                dump[P.rank_col[r]] = k == 2 ? a.shorten_llr : 0.0;
    if (given)
    {
        const double *in = a.llr_in + frame * nc;
        for (int s = tid; s < n_stage; s += kThreads)
        {
            const double L = in[s];
            if (dump)
                dump[s] = L;
            put(s, L);
        }
    }
    else if (a.mode == kModeAwgn)
    {
        const AwgnFrame f = awgn_frame(a, frame, nct);
        ok = f.ok;
        if (f.ok)
            for (int i = tid; i < n_stage; i += kThreads)
            {
                const uint32_t ui = static_cast<uint32_t>(i);
                const uint32_t off = ui + (ui >= f.t1 ? f.d1 : 0u) + (ui >= f.t2 ? f.d2 : 0u);
                const uint64_t w = __builtin_nontemporal_load(f.p0 + off);
                int xb = 0;
                int col = 0;
                if (cw || dump)
                    col = P.bit_pos[i];
                if (cw)
                    xb = static_cast<int>(cw[col]);
                const double noise = dm_from_bits(w) * a.sigma + 0.0; // channel.cpp:62-68
                const double xs = cw ? static_cast<double>(1 - 2 * xb) : 1.0;
                const double y = noise + xs;
                const double L = dm_div_by(2 * y, a.sigma2, a.inv_sigma2); // (2 y) / sigma^2, channel.cpp:88-92: the IEEE quotient
                if (dump)
                    dump[col] = L;
                put(i, L);
            }
    }
    else // kModeBsc (channel.cpp:129-162)
    {
        const uint64_t *raw = a.raw + frame * static_cast<uint64_t>(nct);
        for (int i = tid; i < n_stage; i += kThreads)
        {
            int xb = 0, col = 0;
            if (cw || dump)
                col = P.bit_pos[i];
            if (cw)
                xb = static_cast<int>(cw[col]);
            const int flip = canonical(raw[i]) < a.eps;
            const int y = xb ^ flip;
            const double L = a.delta * static_cast<double>(1 - 2 * y);
            if (dump)
                dump[col] = L;
            put(i, L);
        }
    }
    if (tid == kThreads - 1 && F.has_shortened && !given)
        put(n_stage, a.shorten_llr);
    return ok;
}

// =======================================================================================================================
template <bool WANT_LLR, int VNB, int CNL, bool EXCL, bool HO = false>
__device__ __forceinline__ void fused_body(const DecodeArgs &a, const DevFusedPlan &F)
{
    extern __shared__ double lds[];
    __shared__ int misc[4];
    __shared__ int votes[2][kDecodeWaves];
    const DevPlan &P = a.plan;
    const int nc = P.nc, nct = P.nct;
    const uint64_t frame = blockIdx.x;
    char *msg = reinterpret_cast<char *>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_amdgcn_s_setprio(LDPC_AMD_DECODE_PRIO); // (kernels.hip: the noise generator's waves share the compute units)
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;
    if (tid == 0)
        misc[0] = 0;
    Track t{DM_BOX_LO_WORD, DM_BOX_LO_WORD, 0u};

    // ---- prologue, part 1: stage entries {lambda, rho} ----
    const bool given = a.mode == kModeLlr;
    const int n_stage = given ? nc : nct;
    double2 *stage = reinterpret_cast<double2 *>(lds);
    {
        // rho_ch = e^L for every value; lambda_ch = e^-L beside it only where a node of degree >= 3 will ask for it (the n = 1024
        // code: nowhere — its wide nodes are punctured): no division in the prologue
        const bool need_lambda = F.need_lambda != 0 || given;
        auto put = [&](int s, double L) {
            if (!(__builtin_fabs(L) <= DM_RATIO_LLR_LIMIT))
                t.hmax = 0xFFFFFFFFu; // the frame leaves the ratio form at once
            stage[s] = double2{need_lambda ? dm_exp_clamped(0.0 - L) : 1.0, dm_exp_clamped(L)};
        };
        if (!stage_channel(a, F, frame, tid, put))
            t.hmax = 0xFFFFFFFFu;
        if (tid == kThreads - 1)
            stage[n_stage + 1] = double2{1.0, 1.0}; // L = 0: punctured, never written by the channel, no node
    }

    // what the lane keeps for the whole decode: slot offsets, and (after the barrier) channel values
    const uint32_t *tab = F.lane_tab + (static_cast<uint32_t>(wave) * kFusedLaneRows) * kWaveSize + lane;
    const auto my_vdesc = uniform_table(F.vn_desc + wave * kFusedVnSlots * 4);
    const uint32_t vn_prog = F.vn_prog[wave];
    auto vn_cnt = [&](int w) { return my_vdesc[4 * w] & 0xFFFFu; };
    auto vn_deg = [&](int w) { return my_vdesc[4 * w] >> 16; };
    // EXCL (the small instantiation; plan: FusedPlan::wide_exclusive): a wave either serves ONE block through register-held
    // slot offsets (degree 3..15, slot 0) or blocks of degree 2, never both — the same eight registers m[] hold the wide
    // block's sixteen offsets or the four packed offset words and two of the four channel ratios (twelve registers for the
    // variable-node state instead of twenty: what six waves per SIMD need)
    static_assert(!EXCL || VNB == 4, "the shared layout is written for four slots");
    const bool is_wide = (vn_prog & 0xFu) == kFusedVnWide; // (wave-uniform)
    uint32_t my_idx[VNB + 1], wide_idx[8], vn_entry[VNB], leaf_entry[2 * CNL];
    if constexpr (EXCL)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wide_idx[i] = tab[((is_wide ? 8 : 0) + i) * kWaveSize];
#pragma unroll
        for (int i = 4; i < 8; ++i)
            wide_idx[i] = tab[(8 + i) * kWaveSize];
    }
    else
    {
#pragma unroll
        for (int w = 0; w < VNB; ++w)
            my_idx[w] = tab[w * kWaveSize];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            wide_idx[i] = tab[(8 + i) * kWaveSize];
    }
    my_idx[VNB] = 0;
    const uint32_t none_entry = static_cast<uint32_t>(n_stage + 1) * 16u;
#pragma unroll
    for (int w = 0; w < VNB; ++w)
    {
        if (given)
        {
            const uint32_t cwd = tab[(24 + w) * kWaveSize];
            vn_entry[w] = cwd == kFusedNone ? none_entry : (cwd & 0x3FFFFFFFu) * 16u;
        }
        else
            vn_entry[w] = tab[(16 + w) * kWaveSize];
    }
#pragma unroll
    for (int c = 0; c < 2 * CNL; ++c)
    {
        if (given)
        {
            const uint32_t cwd = tab[(36 + c) * kWaveSize];
            leaf_entry[c] = cwd == kFusedNone ? none_entry : (cwd & 0x3FFFFFFFu) * 16u;
        }
        else
            leaf_entry[c] = tab[(32 + c) * kWaveSize];
    }
    __syncthreads();

    // ---- prologue, part 2: every lane picks up its values; first v2c messages (decoder.cpp:16-19): rho_ch ----
    double my_val[VNB + 1]; // degree 2: rho_ch; degree >= 3: lambda_ch
    double leaf_rho[2 * CNL];
    double first_v2c[VNB];
    my_val[VNB] = 1.0;
#pragma unroll
    for (int w = 0; w < VNB; ++w)
    {
        const double2 e = *reinterpret_cast<const double2 *>(msg + vn_entry[w]);
        first_v2c[w] = e.y;
        my_val[w] = vn_deg(w) == 2 ? e.y : e.x; // (wave-uniform)
    }
#pragma unroll
    for (int c = 0; c < 2 * CNL; ++c)
        leaf_rho[c] = reinterpret_cast<const double2 *>(msg + leaf_entry[c])->y;
    __syncthreads();
    auto idx_at = [&](int w) { return EXCL ? wide_idx[w] : my_idx[w]; };
#pragma unroll
    for (int w = 0; w < VNB; ++w)
    {
        const int cnt = vn_cnt(w), deg = vn_deg(w);
        if (lane >= cnt)
            continue;
        if (deg == 2)
        {
            *at(msg, idx_at(w) & 0xFFFFu) = first_v2c[w];
            *at(msg, idx_at(w) >> 16) = first_v2c[w];
        }
        else if (w == 0 && deg <= 15)
        {
#pragma unroll
            for (int q = 0; q < 15; ++q)
                if (q < deg)
                    *at(msg, (wide_idx[q >> 1] >> (16 * (q & 1))) & 0xFFFFu) = first_v2c[w];
        }
        else if constexpr (!EXCL)
        {
            const uint32_t *idx = F.vn_slot + my_vdesc[4 * w + 1] + lane;
            for (int p = 0; p < deg; ++p)
                *at(msg, idx[p * cnt]) = first_v2c[w];
        }
    }
    if constexpr (EXCL)
        if (!is_wide)
        {
            wide_idx[4] = static_cast<uint32_t>(dm_bits(my_val[1])), wide_idx[5] = hi_word(my_val[1]);
            wide_idx[6] = static_cast<uint32_t>(dm_bits(my_val[2])), wide_idx[7] = hi_word(my_val[2]);
        }
    auto val_at = [&](int w) {
        if constexpr (EXCL)
        {
            if (w == 1)
                return dm_from_bits(wide_idx[4] | (static_cast<uint64_t>(wide_idx[5]) << 32));
            if (w == 2)
                return dm_from_bits(wide_idx[6] | (static_cast<uint64_t>(wide_idx[7]) << 32));
        }
        return my_val[w];
    };
    __syncthreads();

    const auto my_leaf_calls = uniform_table(reinterpret_cast<const uint32_t *>(F.leaf_calls + wave * kFusedLeafCalls));
    const auto my_calls = uniform_table(reinterpret_cast<const uint32_t *>(F.calls + wave * F.calls_stride));
    double *out_llr = WANT_LLR ? a.llr_out + frame * nc : nullptr;
    uint32_t leaf_bits = 0; // bit 2c + h: decision of this lane's leaf in block h of leaf call c, as of the previous pass
    uint32_t vn_bits = 0;   // bit w: decision of this lane's node in slot w
    [[maybe_unused]] double leaf_tot[2 * CNL];
    uint32_t I = 0;
    uint32_t bad = 0, new_leaf_bits = 0, bits = 0;
    [[maybe_unused]] int32_t ho_key = 0; // HO: running maximum of dm_handover_key over the variable nodes' totals
    auto leaf_calls_now = my_leaf_calls;
    auto vdesc_now = my_vdesc;
    uint32_t prog = vn_prog;
    auto refresh_tables = [&] {
        // (the tables are read again in every pass, through pointers the compiler cannot see through: hoisted out of the loop,
        // the descriptors and every condition derived from them sit in scalar registers for the whole decode — more than
        // there are, and the surplus is spilled to vector-register lanes, v_readlane by v_readlane)
        leaf_calls_now = my_leaf_calls;
        vdesc_now = my_vdesc;
        prog = vn_prog;
        asm volatile("" : "+s"(leaf_calls_now), "+s"(vdesc_now), "+s"(prog));
    };
    auto cn_pass = [&] {
        bad = 0, new_leaf_bits = 0;
#pragma unroll
        for (int c = 0; c < CNL; ++c)
        {
            const uint32_t offs = leaf_calls_now[4 * c], cnts = leaf_calls_now[4 * c + 1], cls = leaf_calls_now[4 * c + 2];
            const uint32_t cnt0 = cnts & 0xFFFFu;
            if (cnt0 == 0)
                continue;
            const bool two = (cnts >> 16) != 0;
            uint32_t lb0 = 0, lb1 = 0;
            const uint32_t prev0 = leaf_bits << (31 - 2 * c), prev1 = leaf_bits << (30 - 2 * c);
            double tot0 = 1.0, tot1 = 1.0;
            uint32_t par = 0;
#define LDPC_LEAF_CLASS(D, NM)                                                                                                                 \
    case cls_key(D, 1, NM):                                                                                                                    \
        if (two)                                                                                                                               \
            par = cnf_call<D, true, cls_flip(D, 1, NM), true, WANT_LLR, !HO>(msg, offs & 0xFFFFu, offs >> 16, cnt0, lane, leaf_rho[2 * c],          \
                                                                        leaf_rho[2 * c + 1], prev0, prev1, lb0, lb1, t, tot0, tot1);           \
        else if (lane < static_cast<int>(cnt0))                                                                                                \
            par = cnf_call<D, true, cls_flip(D, 1, NM), false, WANT_LLR, !HO>(msg, offs & 0xFFFFu, 0, cnt0, lane, leaf_rho[2 * c],                  \
                                                                         leaf_rho[2 * c + 1], prev0, prev1, lb0, lb1, t, tot0, tot1);          \
        break;
            switch (cls) // wave-uniform
            {
                LDPC_LEAF_CLASS(3, 0)
                LDPC_LEAF_CLASS(3, 1)
                LDPC_LEAF_CLASS(3, 2)
                LDPC_LEAF_CLASS(4, 0)
                LDPC_LEAF_CLASS(4, 1)
                LDPC_LEAF_CLASS(4, 2)
                LDPC_LEAF_CLASS(4, 3)
            default: break;
            }
#undef LDPC_LEAF_CLASS
            bad |= par;
            new_leaf_bits |= (lb0 << (2 * c)) | (lb1 << (2 * c + 1));
            if constexpr (WANT_LLR)
                leaf_tot[2 * c] = tot0, leaf_tot[2 * c + 1] = tot1;
        }
        for (int c = 0; c < F.calls_stride; ++c)
        {
            const uint32_t offs = my_calls[4 * c], cnts = my_calls[4 * c + 1], cls = my_calls[4 * c + 2];
            const uint32_t cnt0 = cnts & 0xFFFFu;
            if (cnt0 == 0)
                break;
            const bool two = (cnts >> 16) != 0;
            uint32_t lb0, lb1;
            double tot0, tot1;
            uint32_t par = 0;
#define LDPC_CLASS(D, NM)                                                                                                                      \
    case cls_key(D, 0, NM):                                                                                                                    \
        if (two)                                                                                                                               \
            par = cnf_call<D, false, cls_flip(D, 0, NM), true, false, !HO>(msg, offs & 0xFFFFu, offs >> 16, cnt0, lane, 0.0, 0.0, 0u, 0u, lb0, lb1, \
                                                                      t, tot0, tot1);                                                          \
        else if (lane < static_cast<int>(cnt0))                                                                                                \
            par = cnf_call<D, false, cls_flip(D, 0, NM), false, false, !HO>(msg, offs & 0xFFFFu, 0, cnt0, lane, 0.0, 0.0, 0u, 0u, lb0, lb1, t,      \
                                                                       tot0, tot1);                                                            \
        break;
            switch (cls) // wave-uniform
            {
                LDPC_CLASS(2, 0)
                LDPC_CLASS(2, 1)
                LDPC_CLASS(2, 2)
                LDPC_CLASS(3, 0)
                LDPC_CLASS(3, 1)
                LDPC_CLASS(3, 2)
                LDPC_CLASS(3, 3)
                LDPC_CLASS(4, 0)
                LDPC_CLASS(4, 1)
                LDPC_CLASS(4, 2)
                LDPC_CLASS(4, 3)
                LDPC_CLASS(4, 4)
            default: break;
            }
#undef LDPC_CLASS
            bad |= par;
        }
    };
    auto commit_leaves = [&] {
        leaf_bits = new_leaf_bits; // the leaves' decisions of pass I (decoder.cpp:58 for a node of degree 1)
        if constexpr (WANT_LLR)
        {
#pragma unroll
            for (int c = 0; c < 2 * CNL; ++c)
                if (const uint32_t cwd = tab[(36 + c) * kWaveSize]; cwd != kFusedNone)
                    out_llr[cwd & 0x3FFFFFFFu] = 0.0 - dm_log(leaf_tot[c]);
        }
    };
    auto vn_pass = [&] {
        // ---- variable-node pass I, APP and hard decision: decoder.cpp:48-64 ----
        bits = 0;
        auto note = [&](int w, uint32_t sg) { bits |= (sg >> 31) << w; };
        auto note_total = [&](double total) { // HO: how far the node's total is from 1 (detmath.h, dm_handover_key)
            if constexpr (HO)
            {
                const int32_t k = dm_handover_key(total);
                ho_key = k > ho_key ? k : ho_key;
            }
        };
        [[maybe_unused]] auto put_llr = [&](int w, double llr) {
            if constexpr (WANT_LLR)
                out_llr[tab[(24 + w) * kWaveSize] & 0x3FFFFFFFu] = llr;
        };
        auto one = [&](int w, uint32_t kind) {
            const uint32_t d0 = vdesc_now[4 * w];
            const int cnt = static_cast<int>(d0 & 0xFFFFu), deg = static_cast<int>(d0 >> 16);
            if (lane >= cnt)
                return;
            uint32_t sg;
            if (kind == kFusedVnPair || kind == kFusedVn2)
            {
                double tt;
                vn2_one(msg, idx_at(w), val_at(w), t, sg, tt);
                note_total(tt);
                put_llr(w, dm_log(tt));
            }
            else
            {
                double prod = 1.0;
                if (w == 0 && kind == kFusedVnWide)
                    switch (deg) // wave-uniform
                    {
#define LDPC_VN(DV) \
    case DV: prod = vn_wide<DV, EXCL>(msg, wide_idx, my_val[0], t, sg); break;
                        LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7) LDPC_VN(8) LDPC_VN(9)
                        LDPC_VN(10) LDPC_VN(11) LDPC_VN(12) LDPC_VN(13) LDPC_VN(14) LDPC_VN(15)
#undef LDPC_VN
                    default: sg = 0; break;
                    }
                else if constexpr (!EXCL) // (the small instantiation takes plans without such blocks: FusedPlan::wide_exclusive)
                    prod = vn_table(msg, F.vn_slot + vdesc_now[4 * w + 1], lane, cnt, deg, val_at(w), t, sg);
                else
                    sg = 0;
                note_total(prod);
                put_llr(w, 0.0 - dm_log(prod));
            }
            note(w, sg);
        };
#pragma unroll
        for (int w = 0; w < VNB; w += 2)
        {
            const uint32_t k01 = (prog >> (4 * w)) & 0xFFu; // (wave-uniform)
            if (k01 == (kFusedVnPair | (kFusedVnPair << 4)))
            {
                uint32_t sga, sgb;
                double ta, tb;
                vn2_pair(msg, idx_at(w), idx_at(w + 1), val_at(w), val_at(w + 1), t, sga, sgb, ta, tb);
                note(w, sga), note(w + 1, sgb);
                note_total(ta), note_total(tb);
                put_llr(w, dm_log(ta)), put_llr(w + 1, dm_log(tb));
            }
            else
            {
                if (k01 & 0xFu)
                    one(w, k01 & 0xFu);
                if (k01 >> 4)
                    one(w + 1, k01 >> 4);
            }
        }
        vn_bits = bits;
    };
    if constexpr (HO)
    {
        // Without early termination (detmath.h "Hand-over", separately divided check-node outputs): the checks of a pass come
        // FIRST — on what the prologue or the variable-node pass before left behind, voted at the barrier that ends it — so a
        // frame that has run its iterations makes no check-node pass too many, and the pass that hands a frame over can be
        // the one that writes its c2v messages, as LLRs in the general plan's slot order, for the LLR-domain kernel
        // (kernels.hip, decode_kernel<..., RATIO = false> resuming) to continue with.
        auto post_vote = [&](uint32_t pass) {
            const int wave_vote = ((__ballot(dm_box_escaped(t.hmax, t.hmin)) != 0) << 1) | ((__ballot(DM_HANDOVER_DUE(ho_key)) != 0) << 2);
            if (lane == 0)
                votes[pass & 1][wave] = wave_vote;
        };
        post_vote(0);
        __syncthreads();
        for (;;)
        {
            int any = 0;
#pragma unroll
            for (int w = 0; w < kDecodeWaves; ++w)
                any |= votes[I & 1][w];
            if (any & 2) // a value left the box: the LLR-domain form decodes the frame from scratch
            {
                if (tid == 0)
                {
                    const uint32_t pos = atomicAdd(a.redo_count, 1u);
                    a.redo_list[pos] = static_cast<uint32_t>(frame);
                    a.redo_iter[pos] = 0xFFFFFFFFu;
                }
                return;
            }
            if (I == a.iterations)
                break;
            refresh_tables();
            if (any & 4) // a total of variable-node pass I-1 left the inner box: check-node pass I is the last one here
            {
                if (tid == 0)
                {
                    const uint32_t pos = atomicAdd(a.redo_count, 1u);
                    a.redo_list[pos] = static_cast<uint32_t>(frame);
                    a.redo_iter[pos] = I;
                    misc[1] = static_cast<int>(pos);
                }
                __syncthreads();
                double *dst = a.ws_handover + static_cast<uint64_t>(static_cast<uint32_t>(misc[1])) * P.nnz;
                // every output of every check node as lambda, divided on its own, leaves included (dm_cnf*(flip 0, leaf 0,
                // shared 0)); c2v = -log lambda.  Once per frame: rolled, no templates.
                auto ho_call = [&](uint32_t offs, uint32_t cnts, uint32_t cls, int leaf_slot) {
                    const int D = static_cast<int>(cls & 7u), leaf = static_cast<int>((cls >> 3) & 1u), M = D - leaf;
                    for (int h = 0; h < 2; ++h)
                    {
                        const uint32_t off = h ? offs >> 16 : offs & 0xFFFFu, cnt = h ? cnts >> 16 : cnts & 0xFFFFu;
                        if (static_cast<uint32_t>(lane) >= cnt)
                            continue;
                        double v[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < M)
                                v[k] = __builtin_fabs(*at(msg, off + (k * cnt + lane) * 8u));
                        const double rl = leaf ? leaf_rho[(2 * leaf_slot + h) & (2 * CNL - 1)] : 1.0;
                        uint32_t lb = 0;
                        if (D == 2)
                            dm_cnf2(v, 0u);
                        else if (D == 3)
                        {
                            if (leaf)
                                v[2] = rl;
                            dm_cnf3(v, 0u, 0, 0, &lb, nullptr);
                        }
                        else
                        {
                            if (leaf)
                                v[3] = rl;
                            dm_cnf4(v, 0u, 0, 0, &lb, nullptr);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < M)
                                dst[F.ho_map[off / 8u + k * cnt + lane]] = 0.0 - dm_log(v[k]);
                        if (leaf)
                            dst[tab[(40 + ((2 * leaf_slot + h) & 3)) * kWaveSize]] = 0.0 - dm_log(D == 3 ? v[2] : v[3]);
                    }
                };
                for (int c = 0; c < CNL; ++c)
                    if ((leaf_calls_now[4 * c + 1] & 0xFFFFu) != 0)
                        ho_call(leaf_calls_now[4 * c], leaf_calls_now[4 * c + 1], leaf_calls_now[4 * c + 2], c);
                for (int c = 0; c < F.calls_stride; ++c)
                {
                    if ((my_calls[4 * c + 1] & 0xFFFFu) == 0)
                        break;
                    ho_call(my_calls[4 * c], my_calls[4 * c + 1], my_calls[4 * c + 2], 0);
                }
                return;
            }
            cn_pass();
            commit_leaves();
            __syncthreads();
            vn_pass();
            post_vote(I + 1);
            __syncthreads();
            ++I;
        }
    }
    else
    for (;;)
    {
        // ---- loop pass I: check-node pass I, which also sees — in the sign bits of its inputs and the leaf bits — the
        // syndrome of the decisions of variable-node pass I-1 (decoder.cpp:25-45, decoder.h:47-64) ----
        refresh_tables();
        cn_pass();
        const int ph = I & 1;
        const bool esc = dm_box_escaped(t.hmax, t.hmin) || t.pmax >= DM_FUSED_P_HI;
        const int wave_vote = (__ballot((bad & 0x80000000u) != 0) != 0) | ((__ballot(esc) != 0) << 1);
        if (lane == 0)
            votes[ph][wave] = wave_vote;
        __syncthreads();
        int any = 0;
#pragma unroll
        for (int w = 0; w < kDecodeWaves; ++w)
            any |= votes[ph][w];
        if (any & 2) // checked before the syndrome: an escaped frame's hard decisions mean nothing
        {
            if (tid == 0)
            {
                const uint32_t pos = atomicAdd(a.redo_count, 1u);
                a.redo_list[pos] = static_cast<uint32_t>(frame);
            }
            return;
        }
        if (I > 0 && a.early_term && !(any & 1)) // decoder.cpp:66-72 after variable-node pass I-1
        {
            --I;
            break;
        }
        if (I == a.iterations)
            break;
        commit_leaves();

        vn_pass();
        __syncthreads();
        ++I;
    }

    // ---- outputs: iteration count (decoder.cpp:74-77), hard decisions, bit errors (ldpcsim.cpp:184-188) ----
    if (tid == 0 && a.iters)
        a.iters[frame] = I;
    uint8_t *hard = a.hard ? a.hard + frame * nc : nullptr;
    int err = 0;
    if (hard || a.bit_errors)
    {
        // (the lane's row of the table is addressed afresh: held since the prologue the pointer costs the loop two registers)
        int lane_now = lane;
        asm volatile("" : "+v"(lane_now));
        const uint32_t *tab = F.lane_tab + (static_cast<uint32_t>(wave) * kFusedLaneRows) * kWaveSize + lane_now;
        auto account = [&](uint32_t cwd, uint32_t bit) {
            if (cwd == kFusedNone)
                return;
            const uint32_t col = cwd & 0x3FFFFFFFu;
            if (hard)
                hard[col] = static_cast<uint8_t>(bit);
            if (cwd & kFusedCounted)
                err += static_cast<int>(bit) != (cw ? static_cast<int>(cw[col]) : 0);
        };
#pragma unroll
        for (int w = 0; w < VNB; ++w)
            account(tab[(24 + w) * kWaveSize], (vn_bits >> w) & 1u);
#pragma unroll
        for (int c = 0; c < 2 * CNL; ++c)
            account(tab[(36 + c) * kWaveSize], (leaf_bits >> c) & 1u);
    }
    if (a.bit_errors)
    {
        err = wave_sum(err);
        if (lane == 0 && err)
            atomicAdd(&misc[0], err);
        __syncthreads();
        if (tid == 0)
            a.bit_errors[frame] = static_cast<uint32_t>(misc[0]);
    }
}

// =======================================================================================================================
// Min-sum (BP_MS) WITHOUT early termination on the same plan (BASELINE.json configs[2]).  The kernel is bound by the LDS:
// four accesses per edge and iteration, the stores at a third of the loads' rate.  Here the 512 leaf edges of the n = 1024
// code have no message slot at all (their check node's lane keeps the leaf's input LLR and its v2c message — the
// reference's (L_ch + c2v) - c2v, decoder.cpp:58-62, which floating point does not make L_ch — in registers), the hard
// decisions stay in registers, and the prologue is the balanced one.  Every operation is the reference's: sign * sign *
// min (decoder.h:17-20: exact, so the order of a check node's inputs does not matter), the variable node's sum in column
// file order, v2c = out - c2v.  LLR-out doubles == the reference's.
template <int D, bool LEAF, bool TWO>
__device__ __forceinline__ void cnms_call(char *msg, uint32_t off0, uint32_t off1, uint32_t cnt0, int lane, double &lv0, double &lv1,
                                          double L0, double L1, double &out0, double &out1)
{
    constexpr int M = D - (LEAF ? 1 : 0);
    const uint32_t stride = TWO ? kWaveSize * 8u : cnt0 * 8u;
    char *p0 = msg + off0 + lane * 8, *p1 = msg + off1 + lane * 8;
    double v0[D], v1[D];
#pragma unroll
    for (int k = 0; k < M; ++k)
        v0[k] = *at(p0, k * stride);
    if constexpr (TWO)
    {
#pragma unroll
        for (int k = 0; k < M; ++k)
            v1[k] = *at(p1, k * stride);
    }
    if constexpr (LEAF)
        v0[M] = lv0, v1[M] = lv1;
    cn_core<D, true>(v0);
    if constexpr (TWO)
        cn_core<D, true>(v1);
#pragma unroll
    for (int k = 0; k < M; ++k)
        *at(p0, k * stride) = v0[k];
    if constexpr (TWO)
    {
#pragma unroll
        for (int k = 0; k < M; ++k)
            *at(p1, k * stride) = v1[k];
    }
    if constexpr (LEAF) // the leaf's variable-node update (decoder.cpp:50-62 for a node of degree 1), by this lane
    {
        out0 = L0 + v0[M];
        lv0 = out0 - v0[M];
        if constexpr (TWO)
        {
            out1 = L1 + v1[M];
            lv1 = out1 - v1[M];
        }
    }
}

__device__ __forceinline__ void vnms2_pair(char *msg, uint32_t packed_a, uint32_t packed_b, double La, double Lb, double &oa, double &ob)
{
    asm volatile("" : "+v"(packed_a), "+v"(packed_b)); // unpack here, every pass
    const uint32_t a0 = packed_a & 0xFFFFu, a1 = packed_a >> 16, b0 = packed_b & 0xFFFFu, b1 = packed_b >> 16;
    const double ca0 = *at(msg, a0), ca1 = *at(msg, a1), cb0 = *at(msg, b0), cb1 = *at(msg, b1);
    oa = La + ca0, ob = Lb + cb0; // sequential sum in column file order (decoder.cpp:50-56)
    oa += ca1, ob += cb1;
    *at(msg, a0) = oa - ca0, *at(msg, a1) = oa - ca1;
    *at(msg, b0) = ob - cb0, *at(msg, b1) = ob - cb1;
}

__device__ __forceinline__ double vnms2_one(char *msg, uint32_t packed_a, double La)
{
    asm volatile("" : "+v"(packed_a));
    const uint32_t a0 = packed_a & 0xFFFFu, a1 = packed_a >> 16;
    const double ca0 = *at(msg, a0), ca1 = *at(msg, a1);
    double oa = La + ca0;
    oa += ca1;
    *at(msg, a0) = oa - ca0, *at(msg, a1) = oa - ca1;
    return oa;
}

template <int DV>
__device__ __forceinline__ double vnms_wide(char *msg, const uint32_t (&packed)[8], double L)
{
    uint32_t pk[(DV + 1) / 2];
    auto opaque = [&] {
#pragma unroll
        for (int i = 0; i < (DV + 1) / 2; ++i)
        {
            pk[i] = packed[i];
            asm volatile("" : "+v"(pk[i]));
        }
    };
    auto slot = [&](int p) { return at(msg, (p & 1) ? pk[p >> 1] >> 16 : pk[p >> 1] & 0xFFFFu); };
    opaque();
    double c[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c[p] = *slot(p);
    double out = L;
#pragma unroll
    for (int p = 0; p < DV; ++p)
        out += c[p];
    opaque();
#pragma unroll
    for (int p = 0; p < DV; ++p)
        *slot(p) = out - c[p];
    return out;
}

__device__ __forceinline__ double vnms_table(char *msg, const uint32_t *tbl, int lane, int count, int degree, double L)
{
    double out = L;
    for (int p = 0; p < degree; ++p)
        out += *at(msg, tbl[p * count + lane]);
    for (int p = 0; p < degree; ++p)
    {
        double *m = at(msg, tbl[p * count + lane]);
        *m = out - *m;
    }
    return out;
}

template <bool WANT_LLR, int VNB, int CNL>
__device__ __forceinline__ void fused_ms_body(const DecodeArgs &a, const DevFusedPlan &F)
{
    extern __shared__ double lds[];
    __shared__ int misc[4];
    const DevPlan &P = a.plan;
    const int nc = P.nc, nct = P.nct;
    const uint64_t frame = blockIdx.x;
    char *msg = reinterpret_cast<char *>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_amdgcn_s_setprio(LDPC_AMD_DECODE_PRIO);
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;
    if (tid == 0)
        misc[0] = 0;

    // ---- prologue: the input LLRs, staged once (16-byte entries: the lane table's offsets are those of the ratio form) ----
    const bool given = a.mode == kModeLlr;
    const int n_stage = given ? nc : nct;
    double2 *stage = reinterpret_cast<double2 *>(lds);
    bool addressed = stage_channel(a, F, frame, tid, [&](int s, double L) { stage[s].x = L; });
    if (tid == kThreads - 1)
        stage[n_stage + 1].x = 0.0; // punctured, never written by the channel, no node
    const uint32_t *tab = F.lane_tab + (static_cast<uint32_t>(wave) * kFusedLaneRows) * kWaveSize + lane;
    const auto my_vdesc = uniform_table(F.vn_desc + wave * kFusedVnSlots * 4);
    const uint32_t vn_prog = F.vn_prog[wave];
    uint32_t my_idx[VNB], wide_idx[8], vn_entry[VNB], leaf_entry[2 * CNL];
#pragma unroll
    for (int w = 0; w < VNB; ++w)
        my_idx[w] = tab[w * kWaveSize];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        wide_idx[i] = tab[(8 + i) * kWaveSize];
    const uint32_t none_entry = static_cast<uint32_t>(n_stage + 1) * 16u;
    auto entry_of = [&](int mode_row, int col_row) {
        if (given)
        {
            const uint32_t cwd = tab[col_row * kWaveSize];
            return cwd == kFusedNone ? none_entry : (cwd & 0x3FFFFFFFu) * 16u;
        }
        return tab[mode_row * kWaveSize];
    };
#pragma unroll
    for (int w = 0; w < VNB; ++w)
        vn_entry[w] = entry_of(16 + w, 24 + w);
#pragma unroll
    for (int c = 0; c < 2 * CNL; ++c)
        leaf_entry[c] = entry_of(32 + c, 36 + c);
    __syncthreads();
    double my_L[VNB], leaf_L[2 * CNL], leaf_v2c[2 * CNL];
#pragma unroll
    for (int w = 0; w < VNB; ++w)
        my_L[w] = *at(msg, vn_entry[w]);
#pragma unroll
    for (int c = 0; c < 2 * CNL; ++c)
        leaf_L[c] = leaf_v2c[c] = *at(msg, leaf_entry[c]);
    __syncthreads();
    // v2c initialisation: decoder.cpp:16-19
#pragma unroll
    for (int w = 0; w < VNB; ++w)
    {
        const uint32_t d0 = my_vdesc[4 * w];
        const int cnt = static_cast<int>(d0 & 0xFFFFu), deg = static_cast<int>(d0 >> 16);
        if (lane >= cnt)
            continue;
        if (deg == 2)
            *at(msg, my_idx[w] & 0xFFFFu) = my_L[w], *at(msg, my_idx[w] >> 16) = my_L[w];
        else if (w == 0 && deg <= 15)
        {
#pragma unroll
            for (int q = 0; q < 15; ++q)
                if (q < deg)
                    *at(msg, (wide_idx[q >> 1] >> (16 * (q & 1))) & 0xFFFFu) = my_L[w];
        }
        else
        {
            const uint32_t *idx = F.vn_slot + my_vdesc[4 * w + 1];
            for (int p = 0; p < deg; ++p)
                *at(msg, idx[p * cnt + lane]) = my_L[w];
        }
    }
    __syncthreads();

    const auto my_leaf_calls = uniform_table(reinterpret_cast<const uint32_t *>(F.leaf_calls + wave * kFusedLeafCalls));
    const auto my_calls = uniform_table(reinterpret_cast<const uint32_t *>(F.calls + wave * F.calls_stride));
    double *out_llr = WANT_LLR ? a.llr_out + frame * nc : nullptr;
    uint32_t leaf_bits = 0, vn_bits = 0;
    for (uint32_t I = 0; I < a.iterations; ++I)
    {
        const bool last = I + 1 == a.iterations;
        auto leaf_calls_now = my_leaf_calls;
        auto vdesc_now = my_vdesc;
        uint32_t prog = vn_prog;
        asm volatile("" : "+s"(leaf_calls_now), "+s"(vdesc_now), "+s"(prog)); // (see fused_body)
        // ---- check-node pass: decoder.cpp:25-45 ----
        uint32_t lbits = 0;
#pragma unroll
        for (int c = 0; c < CNL; ++c)
        {
            const uint32_t offs = leaf_calls_now[4 * c], cnts = leaf_calls_now[4 * c + 1], cls = leaf_calls_now[4 * c + 2];
            const uint32_t cnt0 = cnts & 0xFFFFu;
            if (cnt0 == 0)
                continue;
            const bool two = (cnts >> 16) != 0;
            double o0 = 1.0, o1 = 1.0;
#define LDPC_MS_LEAF(D)                                                                                                                   \
    case D:                                                                                                                               \
        if (two)                                                                                                                          \
            cnms_call<D, true, true>(msg, offs & 0xFFFFu, offs >> 16, cnt0, lane, leaf_v2c[2 * c], leaf_v2c[2 * c + 1], leaf_L[2 * c],    \
                                     leaf_L[2 * c + 1], o0, o1);                                                                          \
        else if (lane < static_cast<int>(cnt0))                                                                                           \
            cnms_call<D, true, false>(msg, offs & 0xFFFFu, 0, cnt0, lane, leaf_v2c[2 * c], leaf_v2c[2 * c + 1], leaf_L[2 * c],            \
                                      leaf_L[2 * c + 1], o0, o1);                                                                         \
        break;
            switch (cls & 7u) // wave-uniform (a min-sum node does not care where its outputs go: degree and leaf only)
            {
                LDPC_MS_LEAF(3)
                LDPC_MS_LEAF(4)
            default: break;
            }
#undef LDPC_MS_LEAF
            lbits |= (o0 <= 0 ? 1u : 0u) << (2 * c) | (o1 <= 0 ? 1u : 0u) << (2 * c + 1);
            if constexpr (WANT_LLR)
                if (last)
                {
                    if (const uint32_t cwd = tab[(36 + 2 * c) * kWaveSize]; cwd != kFusedNone)
                        out_llr[cwd & 0x3FFFFFFFu] = o0;
                    if (const uint32_t cwd = tab[(37 + 2 * c) * kWaveSize]; cwd != kFusedNone)
                        out_llr[cwd & 0x3FFFFFFFu] = o1;
                }
        }
        leaf_bits = lbits;
        for (int c = 0; c < F.calls_stride; ++c)
        {
            const uint32_t offs = my_calls[4 * c], cnts = my_calls[4 * c + 1], cls = my_calls[4 * c + 2];
            const uint32_t cnt0 = cnts & 0xFFFFu;
            if (cnt0 == 0)
                break;
            const bool two = (cnts >> 16) != 0;
            double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#define LDPC_MS(D)                                                                                                        \
    case D:                                                                                                               \
        if (two)                                                                                                          \
            cnms_call<D, false, true>(msg, offs & 0xFFFFu, offs >> 16, cnt0, lane, d0, d1, 0.0, 0.0, d2, d3);             \
        else if (lane < static_cast<int>(cnt0))                                                                           \
            cnms_call<D, false, false>(msg, offs & 0xFFFFu, 0, cnt0, lane, d0, d1, 0.0, 0.0, d2, d3);                     \
        break;
            switch (cls & 7u)
            {
                LDPC_MS(2)
                LDPC_MS(3)
                LDPC_MS(4)
            default: break;
            }
#undef LDPC_MS
        }
        __syncthreads();

        // ---- variable-node pass, APP and hard decision: decoder.cpp:48-64 ----
        uint32_t bits = 0;
        [[maybe_unused]] auto put_llr = [&](int w, double llr) {
            if constexpr (WANT_LLR)
                if (last)
                    out_llr[tab[(24 + w) * kWaveSize] & 0x3FFFFFFFu] = llr;
        };
        auto one = [&](int w, uint32_t kind) {
            const uint32_t d0 = vdesc_now[4 * w];
            const int cnt = static_cast<int>(d0 & 0xFFFFu), deg = static_cast<int>(d0 >> 16);
            if (lane >= cnt)
                return;
            double out = 1.0;
            if (kind == kFusedVnPair || kind == kFusedVn2)
                out = vnms2_one(msg, my_idx[w], my_L[w]);
            else if (w == 0 && kind == kFusedVnWide)
                switch (deg) // wave-uniform
                {
#define LDPC_VN(DV) \
    case DV: out = vnms_wide<DV>(msg, wide_idx, my_L[0]); break;
                    LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7) LDPC_VN(8) LDPC_VN(9)
                    LDPC_VN(10) LDPC_VN(11) LDPC_VN(12) LDPC_VN(13) LDPC_VN(14) LDPC_VN(15)
#undef LDPC_VN
                default: break;
                }
            else
                out = vnms_table(msg, F.vn_slot + vdesc_now[4 * w + 1], lane, cnt, deg, my_L[w]);
            bits |= (out <= 0 ? 1u : 0u) << w;
            put_llr(w, out);
        };
#pragma unroll
        for (int w = 0; w < VNB; w += 2)
        {
            const uint32_t k01 = (prog >> (4 * w)) & 0xFFu; // (wave-uniform)
            if (w + 1 < VNB && k01 == (kFusedVnPair | (kFusedVnPair << 4)))
            {
                double oa, ob;
                vnms2_pair(msg, my_idx[w], my_idx[w + 1], my_L[w], my_L[w + 1], oa, ob);
                bits |= (oa <= 0 ? 1u : 0u) << w | (ob <= 0 ? 1u : 0u) << (w + 1);
                put_llr(w, oa), put_llr(w + 1, ob);
            }
            else
            {
                if (k01 & 0xFu)
                    one(w, k01 & 0xFu);
                if (w + 1 < VNB && (k01 >> 4))
                    one(w + 1, k01 >> 4);
            }
        }
        vn_bits = bits;
        __syncthreads();
    }

    // ---- outputs: iteration count (decoder.cpp:74-77: no early termination), hard decisions, bit errors (ldpcsim.cpp:184-188) ----
    if (tid == 0 && a.iters)
        a.iters[frame] = a.iterations;
    uint8_t *hard = a.hard ? a.hard + frame * nc : nullptr;
    int err = 0;
    if (hard || a.bit_errors)
    {
        auto account = [&](uint32_t cwd, uint32_t bit) {
            if (cwd == kFusedNone)
                return;
            const uint32_t col = cwd & 0x3FFFFFFFu;
            if (hard)
                hard[col] = static_cast<uint8_t>(bit);
            if (cwd & kFusedCounted)
                err += static_cast<int>(bit) != (cw ? static_cast<int>(cw[col]) : 0);
        };
#pragma unroll
        for (int w = 0; w < VNB; ++w)
            account(tab[(24 + w) * kWaveSize], (vn_bits >> w) & 1u);
#pragma unroll
        for (int c = 0; c < 2 * CNL; ++c)
            account(tab[(36 + c) * kWaveSize], (leaf_bits >> c) & 1u);
    }
    if (a.bit_errors)
    {
        err = wave_sum(err);
        if (lane == 0 && err)
            atomicAdd(&misc[0], err);
        __syncthreads();
        if (tid == 0)
            a.bit_errors[frame] = addressed ? static_cast<uint32_t>(misc[0]) : 0xFFFFFFFFu;
    }
}

template <bool WANT_LLR, int VNB, int CNL>
__global__ __launch_bounds__(kThreads) void decode_fused_ms_kernel(const DecodeArgs a, const DevFusedPlan f)
{
    fused_ms_body<WANT_LLR, VNB, CNL>(a, f);
}

// the instantiation of the n = 1024 code (at most four variable-node blocks and one leaf call per wave, no LLR output) is
// compiled for LDPC_AMD_FUSED_WAVES waves per SIMD = that many frames per CU (its messages take 23 KB of LDS)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(LDPC_AMD_FUSED_WAVES, LDPC_AMD_FUSED_WAVES))) void
decode_fused_small(const DecodeArgs a, const DevFusedPlan f)
{
    fused_body<false, 4, 1, true>(a, f);
}

template <bool WANT_LLR, int VNB, int CNL>
__global__ __launch_bounds__(kThreads) void decode_fused_ho_kernel(const DecodeArgs a, const DevFusedPlan f)
{
    fused_body<WANT_LLR, VNB, CNL, false, true>(a, f);
}

// (no LLR output, four blocks, one leaf call: the n = 1024 code without early termination; separately divided outputs keep
// more values live than the shared reciprocals do: five waves per SIMD — at six the compiler spills 104 bytes per lane)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5, 5))) void
decode_fused_ho_small(const DecodeArgs a, const DevFusedPlan f)
{
    fused_body<false, 4, 1, true, true>(a, f);
}

template <bool WANT_LLR, int VNB, int CNL>
__global__ __launch_bounds__(kThreads) void decode_fused_kernel(const DecodeArgs a, const DevFusedPlan f)
{
    fused_body<WANT_LLR, VNB, CNL, false>(a, f);
}

} // namespace

int launch_decode_fused(const DecodeArgs &a, const DevFusedPlan &f, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    if (!a.redo_list || !a.redo_count || a.redo_count_in || !a.early_term || a.iterations == 0 || a.ratio_separate)
        return hipErrorInvalidValue;
    if (f.vnb > kFusedVnSlots || f.cnl > kFusedLeafCalls)
        return hipErrorInvalidValue;
    const bool want_llr = a.llr_out != nullptr;
    void (*k)(const DecodeArgs, const DevFusedPlan) = nullptr;
    if (f.vnb <= 4 && f.cnl <= 1)
        k = (want_llr || !f.wide_exclusive) ? (want_llr ? decode_fused_kernel<true, 4, 1> : decode_fused_kernel<false, 4, 1>) : decode_fused_small;
    else
        k = want_llr ? decode_fused_kernel<true, kFusedVnSlots, kFusedLeafCalls> : decode_fused_kernel<false, kFusedVnSlots, kFusedLeafCalls>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(f.lds_bytes));
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(a.n_frames)), dim3(kThreads), f.lds_bytes, static_cast<hipStream_t>(stream), a, f);
    return hipGetLastError();
}

int launch_decode_fused_handover(const DecodeArgs &a, const DevFusedPlan &f, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    if (!a.redo_list || !a.redo_count || !a.redo_iter || !a.ws_handover || a.redo_count_in || a.early_term || a.iterations == 0)
        return hipErrorInvalidValue;
    if (f.vnb > kFusedVnSlots || f.cnl > kFusedLeafCalls)
        return hipErrorInvalidValue;
    const bool want_llr = a.llr_out != nullptr;
    void (*k)(const DecodeArgs, const DevFusedPlan) = nullptr;
    if (f.vnb <= 4 && f.cnl <= 1)
        k = (want_llr || !f.wide_exclusive) ? (want_llr ? decode_fused_ho_kernel<true, 4, 1> : decode_fused_ho_kernel<false, 4, 1>) : decode_fused_ho_small;
    else
        k = want_llr ? decode_fused_ho_kernel<true, kFusedVnSlots, kFusedLeafCalls> : decode_fused_ho_kernel<false, kFusedVnSlots, kFusedLeafCalls>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(f.lds_bytes));
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(a.n_frames)), dim3(kThreads), f.lds_bytes, static_cast<hipStream_t>(stream), a, f);
    return hipGetLastError();
}

int launch_decode_fused_minsum(const DecodeArgs &a, const DevFusedPlan &f, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    if (a.early_term || a.iterations == 0 || a.redo_list || a.redo_count_in)
        return hipErrorInvalidValue;
    if (f.vnb > kFusedVnSlots || f.cnl > kFusedLeafCalls)
        return hipErrorInvalidValue;
    const bool want_llr = a.llr_out != nullptr;
    void (*k)(const DecodeArgs, const DevFusedPlan) = nullptr;
    if (f.vnb <= 4 && f.cnl <= 1)
        k = want_llr ? decode_fused_ms_kernel<true, 4, 1> : decode_fused_ms_kernel<false, 4, 1>;
    else
        k = want_llr ? decode_fused_ms_kernel<true, kFusedVnSlots, kFusedLeafCalls> : decode_fused_ms_kernel<false, kFusedVnSlots, kFusedLeafCalls>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(f.lds_bytes));
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(a.n_frames)), dim3(kThreads), f.lds_bytes, static_cast<hipStream_t>(stream), a, f);
    return hipGetLastError();
}

} // namespace ldpc_amd
