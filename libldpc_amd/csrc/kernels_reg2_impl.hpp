// kernels_reg2_impl.hpp — register-resident BP / min-sum decoder, second form (BASELINE config 4: (3,6)-regular
// n=8192, 196 KB of fp64 messages per frame: more than one CU's LDS).
//
// One workgroup of NT threads (1024 or 512) decodes one frame on one CU.  Thread (wave, lane) owns the check nodes of CN blocks
// k*16 + wave (k < KC) and holds their KC x MAXD messages in registers for the whole decode, together with — packed in
// one word per edge — where the edge's c2v message goes in the LDS mailbox and where the total of the edge's variable
// node comes back from.  What differs from kernels_reg.hip (which stays as the general fallback):
//
//   * the edge's owner forms the v2c message itself.  decoder.cpp:50-64 computes  out = LLRin + sum of c2v  and then
//     v2c_e = out - c2v_e  per edge; the subtraction (ratio form: the product  rho(total) * lambda(c2v_e) ) needs only the
//     node's total and the edge's own c2v, which its owner still has in a register.  So the variable-node thread sends
//     back ONE 8-byte total per node instead of a message and a hard-decision byte per edge, and the decision rides in
//     the total (its sign in the ratio form, out <= 0 in the LLR domain): no hard-bit array, the syndrome
//     (decoder.h:47-64) is an XOR over the totals a check node gathers anyway;
//   * input LLRs live in the registers of the variable-node threads (NV0 + NV1 nodes per thread), no device memory
//     in the iteration loop;
//   * two mailbox rounds per iteration, five barriers: gather + vote | CN pass + scatter round 0 | VN round 0 |
//     scatter round 1 | VN round 1;
//   * check nodes of degree >= 5 keep their partial results as undivided fractions (detmath.h, dm_frac).
//
// Same arithmetic, same order as the LDS-resident kernel (kernels.hip) and as the reference:
//   decode loop src/decoding/decoder.cpp:11-78, CN recursion :31-44 (device_cn.hpp), VN sum :50-56 in column
//   file order, syndrome src/decoding/decoder.h:47-64, channels src/sim/channel.cpp (device_channel.hpp).
// Included by kernels_reg2.hip (generic instantiations + the dispatcher) and kernels_reg2u.hip (the regular-code
// instantiations): two translation units compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <utility>

#include "device_channel.hpp"
#include "device_cn.hpp"
#include "device_math.hpp"
#include "kernels.hpp"


namespace ldpc_amd
{

namespace
{

// SH6: the first of the three launches — check nodes of degree 6 share reciprocals (detmath.h, dm_cn6_shared); the range
// check of their products joins `escaped`, which is voted on at the top of the NEXT pass: it counts once the frame has gone
// on to the variable-node pass, as the rule says
template <bool MINSUM, bool RATIO, int MAXD, bool SH6>
__device__ __forceinline__ void cn_regs2(double (&m)[MAXD], int degree, uint32_t *escaped)
{
    // wave-uniform degree: one fully unrolled recursion per width
#define LDPC_CASE(D)                                                \
    case D:                                                         \
    {                                                               \
        double v[D];                                                \
        _Pragma("unroll") for (int j = 0; j < D; ++j) v[j] = m[j];  \
        if constexpr (RATIO)                                        \
            cn_ratio<D, false, SH6>(v, nullptr, escaped);           \
        else                                                        \
            cn_core<D, MINSUM>(v);                                  \
        _Pragma("unroll") for (int j = 0; j < D; ++j) m[j] = v[j];  \
        break;                                                      \
    }
    switch (degree)
    {
        LDPC_CASE(2)
        LDPC_CASE(3)
        LDPC_CASE(4)
    default:
        if constexpr (MAXD >= 6)
            switch (degree)
            {
                LDPC_CASE(5)
                LDPC_CASE(6)
            default:
                if constexpr (MAXD >= 8)
                    switch (degree)
                    {
                        LDPC_CASE(7)
                        LDPC_CASE(8)
                    default: break;
                    }
                break;
            }
        break;
    }
#undef LDPC_CASE
}

__device__ __forceinline__ int wave_sum_i2(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// LDS access by absolute byte address: the packed edge words hold addresses relative to the start of the workgroup's
// LDS, which is where the kernel's only LDS object (the dynamic array) begins — checked once per launch below
template <typename T>
__device__ __forceinline__ T __attribute__((address_space(3))) *lds_abs(uint32_t byte_address)
{
    return (T __attribute__((address_space(3))) *)(static_cast<uintptr_t>(byte_address));
}

__device__ __forceinline__ Reg2VnBlock load_vn_block(const Reg2VnBlock *table, uint32_t i) // {u32, u32, u32, u16, u16}
{
    static_assert(sizeof(Reg2VnBlock) == 16, "descriptor layout");
    const auto t = uniform_table(reinterpret_cast<const uint32_t *>(table));
    const uint32_t w0 = t[4 * i], w1 = t[4 * i + 1], w2 = t[4 * i + 2], w3 = t[4 * i + 3];
    return Reg2VnBlock{w0, w1, w2, static_cast<uint16_t>(w3 & 0xFFFFu), static_cast<uint16_t>(w3 >> 16)};
}

// RATIO: the likelihood-ratio form of the sum-product iteration (detmath.h): c2v messages are lambda = e^-L, the
// returned total is rho(total) = 1 / (lambda(L_ch) * prod lambda(c2v)) with the hard decision in its sign bit, the
// owner's v2c is rho(total) * lambda(c2v_e); a frame that leaves the representable box is given up (reg2_frame returns true).
// UCN: every check-node block of the plan has exactly MAXD edges (a regular code): no switch over the degree, a third of
// the code.  UVN: every variable-node block is full and of degree 3 with columns, rests and totals at affine offsets
// (DevReg2Plan::vn_affine): a round is straight-line code — channel terms fetched before the barrier that opens the
// round, all its mailbox reads in flight together, the divisions of its blocks interleaved.
// One frame in one form.  RATIO: the likelihood-ratio form (sum-product with early termination); SH6: its check nodes of
// degree 6 share reciprocals (detmath.h, dm_cn6_shared: the first of the three attempts a frame may need).  Returns true when
// the frame left the range of the form — a value outside the box of the ratio form, or a product of denominators beyond its
// limit — and has to be decoded again from scratch by the next form; nothing of it has been delivered then.
template <bool MINSUM, bool WANT_LLR, int NT, int KC, int MAXD, int NV0, int NV1, bool RATIO, bool SH6, bool UCN, bool UVN>
__device__ __forceinline__ bool reg2_frame(const DecodeArgs &a, const DevReg2Plan &R_arg, const uint64_t frame, const int wave)
{
    static_assert(NV0 % 2 == 0 && NV1 % 2 == 0, "variable-node rounds go two blocks at a time");
    static_assert(!(RATIO && MINSUM), "the ratio form is a sum-product form");
    static_assert(RATIO || !SH6, "shared reciprocals belong to the ratio form");
    constexpr int W = NT / 64, NV = NV0 + NV1;
    extern __shared__ double lds[]; // R_arg.lds_entries doubles, then two vote words
    const DevPlan &P = a.plan;
    const int nc = P.nc;
    uint32_t *vote = reinterpret_cast<uint32_t *>(lds + R_arg.lds_entries);
    // A frame may be decoded up to three times by this workgroup, one form after the other: the tables' addresses and the
    // thread index are taken afresh for every attempt — the compiler would otherwise form the addresses the later attempts
    // need ahead of the first one and keep them (spilled) across it.  (The lane index from the execution mask, the wave index
    // from the caller's scalar register: a copy of threadIdx.x kept for the later attempts is itself such a value.)
    DevReg2Plan R = R_arg;
    asm volatile("" : "+s"(R.edge_w), "+s"(R.cn_deg), "+s"(R.vn_blocks), "+s"(R.vn_rank));
    int lane = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    double *llr = a.ws_llr + frame * nc;
    uint8_t *hard = a.ws_hb + frame * nc;
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;

    channel_init<NT>(a, frame, llr, tid);
    if (tid == 0)
    {
        lds[R.neutral] = 1.0;
        vote[0] = 0, vote[1] = 0, vote[2] = 0;
    }
    __syncthreads();
    if (a.llr_in_dump)
    {
        double *o = a.llr_in_dump + frame * nc;
        for (int r = tid; r < nc; r += NT)
            o[P.rank_col[r]] = llr[r];
    }

    // ---- variable-node side: input LLRs (RATIO: as lambda = e^-L) into registers, first totals into LDS ----
    // v2c initialisation (decoder.cpp:16-19): every edge starts with its VN's input LLR; with c2v = 0 (lambda = 1) in
    // the owners' registers the first gather yields exactly that.
    // The channel term of a node (its LLR, or lambda = e^-L) is needed once per iteration, by one thread: it lives in
    // device memory, a.ws_scr[frame][i][tid] — 64 KB per frame that stays in the L2 of the XCD — and not in 16 registers
    // the check-node pass needs.
    uint32_t escaped = 0; // RATIO: running maximum of dm_ratio_key over the checked values (detmath.h)
    double *const lam_ws = a.ws_scr + (frame * NV) * NT; // wave-uniform base: [i][tid]
#pragma unroll
    for (int i = 0; i < NV; ++i)
    {
        const Reg2VnBlock vb = load_vn_block(R.vn_blocks, i * W + wave);
        double lam_i = RATIO ? 1.0 : 0.0;
        if (lane < vb.count)
        {
            const double L = llr[R.vn_rank[(i * W + wave) * 64 + lane]];
            if constexpr (RATIO)
            {
                if (!(__builtin_fabs(L) <= DM_RATIO_LLR_LIMIT))
                    escaped = ~0u;
                lam_i = dm_exp_clamped(0.0 - L);
                lds[vb.tot_off + lane] = dm_ratio_div(1.0, lam_i);
            }
            else
            {
                lam_i = L;
                lds[vb.tot_off + lane] = L;
            }
        }
        lam_ws[i * NT + tid] = lam_i;
    }

    // ---- check-node side ----
    double m[KC][MAXD];
    uint32_t ew[KC][MAXD];
    int deg[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k)
    {
        deg[k] = R.cn_deg[k * W + wave];
#pragma unroll
        for (int j = 0; j < MAXD; ++j)
        {
            m[k][j] = RATIO ? 1.0 : 0.0;
            ew[k][j] = R.edge_w[(k * MAXD + j) * NT + tid];
        }
    }
    // every thread writes all its columns in both mailbox rounds; the ones that do not belong to the round (or hold no
    // edge) land in the trash entry (plan.hpp, edge_w): no execution masks, no branches
    const uint32_t trash = kReg2TrashEntry * 8u;
    __syncthreads();

    double *out_llr = WANT_LLR ? a.llr_out + frame * nc : nullptr;
    uint32_t I = 0, result = 0;
#ifdef LDPC_AMD_PHASE_TRACE
    // per-wave cycles spent in each phase of the loop and waiting at its barrier (tools/phase_probe_reg2.py)
    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
    const uint64_t ph_entry = ph_t;
#define REG2_TICK(k)                                          \
    {                                                         \
        const uint64_t n_ = __builtin_amdgcn_s_memtime();     \
        ph[k] += n_ - ph_t;                                   \
        ph_t = n_;                                            \
    }
#else
#define REG2_TICK(k)
#endif
    for (;;)
    {
        // (the same for the two address fields of the packed edge words: hoisted, they would cost 2*KC*MAXD registers)
        uint32_t gather_mask = 0x3FFF8u, scatter_mask = 0x7FFF8u;
        const Reg2VnBlock *vn_blocks = R.vn_blocks;
        asm volatile("" : "+s"(gather_mask), "+s"(scatter_mask), "+s"(vn_blocks));
        // (and for the words themselves: their rotated forms are loop invariants too)
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int j = 0; j < MAXD; ++j)
                asm volatile("" : "+v"(ew[k][j]));
        auto scatter = [&](auto round) {
            // (opaque per call: otherwise round 1 reuses the 24 rotated-and-masked words of round 0, kept in registers
            // across the variable-node round in between)
            asm volatile("" : "+s"(scatter_mask));
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int j = 0; j < MAXD; ++j)
                {
                    const uint32_t at = (__builtin_amdgcn_alignbit(ew[k][j], ew[k][j], 15) & scatter_mask) - decltype(round)::value * 0x20000u;
                    *lds_abs<double>(at < trash ? at : trash) = m[k][j];
                }
        };
        // ---- gather: v2c of every owned edge from its VN's total; syndrome of the decisions in the totals ----
        uint32_t par = 0;
#pragma unroll
        for (int k = 0; k < KC; ++k)
        {
            uint32_t pk = 0;
#pragma unroll
            for (int j = 0; j < MAXD; ++j)
            {
                // (no test on the node's degree here: columns without an edge gather the neutral entry, +1.0, and stay
                // what they are; straight-line code lets all the loads of a check node be in flight together)
                const double t = *lds_abs<const double>(ew[k][j] & gather_mask);
                if constexpr (RATIO)
                {
                    pk ^= DM_SIGN_WORD(t);
                    const double o = __builtin_fabs(t) * m[k][j]; // rho(total - c2v_e)
                    DM_RATIO_TRACK(escaped, o);
                    m[k][j] = o;
                }
                else
                {
                    pk ^= (t <= 0) ? 0x80000000u : 0u;
                    m[k][j] = t - m[k][j];
                }
            }
            par |= pk;
            // keep the loads of the next check node behind this one's arithmetic: all KC*MAXD totals in flight at once
            // would need more registers than the thread has left beside its messages
            asm volatile("" ::: "memory");
        }
        // ---- vote: escaped (ratio form) and syndrome (early termination); also the barrier that lets the scatter
        //      below overwrite the position-0 entries the gather has just read ----
        {
            uint32_t f = (par >> 31) | ((RATIO && DM_RATIO_ESCAPED(escaped)) ? 2u : 0u);
            const uint64_t any1 = __ballot(f & 1u), any2 = __ballot(f & 2u);
            if (lane == 0 && (any1 | any2))
                atomicOr(&vote[I & 1], (any1 ? 1u : 0u) | (any2 ? 2u : 0u));
            REG2_TICK(0)
            __syncthreads();
            REG2_TICK(1)
            const uint32_t v = vote[I & 1];
            if (tid == 0)
                vote[(I + 1) & 1] = 0;
            if (RATIO && (v & 2u)) // checked before the syndrome: an escaped frame's decisions mean nothing
            {
                return true; // (uniform: every thread has read the same vote word)
            }
            if (I > 0 && a.early_term && !(v & 1u)) // decoder.cpp:66-72: the decisions of iteration I-1 are a codeword
            {
                result = I - 1;
                break;
            }
            if (I == a.iterations)
            {
                result = I;
                break;
            }
        }
        // ---- CN pass (decoder.cpp:25-45), entirely in registers; c2v of round-0 edges -> mailbox ----
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
            (([&] {
                 if constexpr (!UCN)
                 {
                     if (deg[Ks] >= 2)
                         cn_regs2<MINSUM, RATIO, MAXD, SH6>(m[Ks], deg[Ks], &escaped);
                 }
                 else if constexpr (RATIO)
                     cn_ratio<MAXD, false, SH6>(m[Ks], nullptr, &escaped);
                 else
                     cn_core<MAXD, MINSUM>(m[Ks]);
                 // one node after the other: the next one's inputs exist when this one's outputs do (left to itself the compiler
                 // works on all KC nodes at once and spills: 204 bytes per lane with the shared reciprocals)
                 if constexpr (UCN && MAXD == 6 && Ks + 1 < KC)
                     asm volatile("" : "+v"(m[Ks][3]), "+v"(m[Ks][4]), "+v"(m[Ks][5]), "+v"(m[Ks + 1][0]), "+v"(m[Ks + 1][1]), "+v"(m[Ks + 1][2]),
                                       "+v"(m[Ks + 1][3]), "+v"(m[Ks + 1][4]), "+v"(m[Ks + 1][5]));
             }()),
             ...);
        }(std::make_integer_sequence<int, KC>{});
        // one node: the total (decoder.cpp:50-56) from its channel term and its column, in column file order
        auto vn_total3 = [&](double l, double c0, double x1, double x2, uint32_t tot_byte, int i) {
            double tot_entry, out_value;
            if constexpr (RATIO)
            {
                double prod = l * c0; // lambda(total) = lambda(L_ch) * prod lambda(c2v_p)
                prod *= x1;
                prod *= x2;
                const uint64_t bit = prod >= 1.0; // total LLR <= 0: the decision rides in the sign of the entry
                const double tot = dm_ratio_div(1.0, prod); // rho(total)
                tot_entry = dm_from_bits(dm_bits(tot) | (bit << 63));
                if constexpr (WANT_LLR)
                    out_value = 0.0 - dm_log(prod);
            }
            else
            {
                double out = l + c0;
                out += x1;
                out += x2;
                tot_entry = out;
                out_value = out;
            }
            *lds_abs<double>(tot_byte) = tot_entry;
            if constexpr (WANT_LLR)
                out_llr[P.rank_col[R.vn_rank[(i * W + wave) * 64 + lane]]] = out_value;
        };
        // UVN: block (round r, i) of this wave sits at entry  base + stride * (i * W + wave) + lane  (DevReg2Plan::vn_affine)
        auto vn_fetch = [&]<int... Is>(std::integer_sequence<int, Is...>, auto base, double(&l)[sizeof...(Is)]) {
            // scalar base + one 32-bit lane offset (opaque: otherwise eight 64-bit lane addresses are hoisted out of the loop)
            uint32_t toff = static_cast<uint32_t>(tid) * 8u;
            asm volatile("" : "+v"(toff));
            ((l[Is] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lam_ws + (decltype(base)::value + Is) * NT) + toff)), ...);
            __builtin_amdgcn_sched_barrier(0); // the loads stay here, ahead of the phase that covers their latency
        };
        auto vn_round_u = [&]<int... Js>(std::integer_sequence<int, Js...>, auto base, auto rnd, const double(&l_in)[sizeof...(Js)],
                                         auto first) {
            // sizeof...(Js) blocks of the round starting at its block `first`, in lock step; arrays below are indexed from 0
            constexpr int r = decltype(rnd)::value, n = sizeof...(Js), j0 = decltype(first)::value;
            [&]<int... Is>(std::integer_sequence<int, Is...>) {
            const double(&l)[n] = l_in;
            uint32_t a0 = R.vn_affine[r][0], s0 = R.vn_affine[r][1], a1 = R.vn_affine[r][2], s1 = R.vn_affine[r][3], a2 = R.vn_affine[r][4],
                     s2 = R.vn_affine[r][5];
            asm volatile("" : "+s"(a0), "+s"(s0), "+s"(a1), "+s"(s1), "+s"(a2), "+s"(s2)); // (not hoisted: 3 registers per block)
            const uint32_t lw = static_cast<uint32_t>(lane) * 8u;
            const uint32_t p0 = (a0 + s0 * wave) * 8u + lw, pr = (a1 + s1 * wave) * 8u + lw, pt = (a2 + s2 * wave) * 8u + lw;
            double c0[n], x1[n], x2[n], prod[n];
            ((c0[Is] = *lds_abs<const double>(p0 + (j0 + Is) * s0 * (W * 8u)), x1[Is] = *lds_abs<const double>(pr + (j0 + Is) * s1 * (W * 8u)),
              x2[Is] = *lds_abs<const double>(pr + (j0 + Is) * s1 * (W * 8u) + 512u)),
             ...);
            __builtin_amdgcn_sched_barrier(0); // every read of the round in flight before the first product waits for one
            if constexpr (RATIO)
            {
                // the blocks in lock step, one operation of each at a time: the dependent chain of a division
                // (dm_ratio_div(1.0, prod), the same instruction sequence) is eight instructions long
                ((prod[Is] = l[Is] * c0[Is]), ...); // lambda(total) = lambda(L_ch) * prod lambda(c2v_p), column file order
                ((prod[Is] *= x1[Is]), ...);
                ((prod[Is] *= x2[Is]), ...);
                double rc[n], e[n];
                ((rc[Is] = __builtin_amdgcn_rcp(prod[Is])), ...);
                ((e[Is] = DM_FMA(-prod[Is], rc[Is], 1.0)), ...);
                ((rc[Is] = DM_FMA(rc[Is], e[Is], rc[Is])), ...);
                ((e[Is] = DM_FMA(-prod[Is], rc[Is], 1.0)), ...);
                ((rc[Is] = DM_FMA(rc[Is], e[Is], rc[Is])), ...);
                ((e[Is] = DM_FMA(-prod[Is], rc[Is], 1.0)), ...); // (the quotient 1.0 * r is r)
                ((e[Is] = DM_FMA(e[Is], rc[Is], rc[Is])), ...);  // rho(total)
                // total LLR <= 0 (prod >= 1): the decision rides in the sign of the entry
                ((*lds_abs<double>(pt + (j0 + Is) * s2 * (W * 8u)) = dm_from_bits(dm_bits(e[Is]) | (static_cast<uint64_t>(prod[Is] >= 1.0) << 63))), ...);
                if constexpr (WANT_LLR)
                    ((out_llr[P.rank_col[R.vn_rank[((decltype(base)::value + j0 + Is) * W + wave) * 64 + lane]]] = 0.0 - dm_log(prod[Is])), ...);
            }
            else
            {
                ((prod[Is] = l[Is] + c0[Is]), ...); // sequential sum in column file order
                ((prod[Is] += x1[Is]), ...);
                ((prod[Is] += x2[Is]), ...);
                ((*lds_abs<double>(pt + (j0 + Is) * s2 * (W * 8u)) = prod[Is]), ...);
                if constexpr (WANT_LLR)
                    ((out_llr[P.rank_col[R.vn_rank[((decltype(base)::value + j0 + Is) * W + wave) * 64 + lane]]] = prod[Is]), ...);
            }
        }(std::make_integer_sequence<int, n>{});
        };
        // a round two blocks at a time: four in lock step would hide more latency and need 20 registers more than the
        // thread has (measured: the spills that buys cost more than the lock step gains)
        auto vn_pairs = [&]<int... Ps>(std::integer_sequence<int, Ps...>, auto base, auto rnd, const double(&l)[2 * sizeof...(Ps)]) {
            (([&] {
                 const double lp[2] = {l[2 * Ps], l[2 * Ps + 1]};
                 vn_round_u(std::integer_sequence<int, 0, 1>{}, base, rnd, lp, std::integral_constant<int, 2 * Ps>{});
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        };
        // ---- VN pass, APP and hard decision (decoder.cpp:48-64): totals of the two rounds ----
        auto vn_round = [&]<int... Is>(std::integer_sequence<int, Is...>, auto base) {
            (([&] {
                 constexpr int i = decltype(base)::value + Is;
                 const Reg2VnBlock vb = load_vn_block(vn_blocks, i * W + wave);
                 uint32_t toff = static_cast<uint32_t>(tid) * 8u;
                 asm volatile("" : "+v"(toff));
                 const double lam_i = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lam_ws + i * NT) + toff);
                 if (lane < vb.count)
                 {
                     const double *c0 = lds + vb.p0_off + lane, *cr = lds + vb.prest_off + lane;
                     if (vb.degree == 3) // unrolled: all loads in flight at once
                     {
                         const double x0 = c0[0], x1 = cr[0], x2 = cr[vb.count];
                         vn_total3(lam_i, x0, x1, x2, (vb.tot_off + lane) * 8u, i);
                         return;
                     }
                     double tot_entry, out_value;
                     if constexpr (RATIO)
                     {
                         // lambda(total) = lambda(L_ch) * prod lambda(c2v_p), in column file order
                         double prod = lam_i * c0[0];
                         if (vb.degree <= 3)
                             for (int p = 1; p < vb.degree; ++p)
                                 prod *= cr[(p - 1) * vb.count];
                         else
                             for (int p = 1; p < vb.degree; ++p)
                             {
                                 prod *= cr[(p - 1) * vb.count];
                                 if (p % 3 == 2)
                                     DM_RATIO_TRACK(escaped, prod);
                             }
                         const uint64_t bit = prod >= 1.0; // total LLR <= 0: the decision rides in the sign of the entry
                         const double tot = dm_ratio_div(1.0, prod); // rho(total)
                         tot_entry = dm_from_bits(dm_bits(tot) | (bit << 63));
                         if constexpr (WANT_LLR)
                             out_value = 0.0 - dm_log(prod);
                     }
                     else
                     {
                         double out = lam_i + c0[0]; // sequential sum in column file order
                         for (int p = 1; p < vb.degree; ++p)
                             out += cr[(p - 1) * vb.count];
                         tot_entry = out;
                         out_value = out;
                     }
                     lds[vb.tot_off + lane] = tot_entry;
                     if constexpr (WANT_LLR)
                         out_llr[P.rank_col[R.vn_rank[(i * W + wave) * 64 + lane]]] = out_value;
                 }
             }()),
             ...);
        };
        if constexpr (UVN)
        {
            // the channel terms of a round come from device memory (L2): asked for a phase ahead, so that the mailbox
            // writes and a barrier cover the latency
            double l0[NV0], l1[NV1];
            vn_fetch(std::make_integer_sequence<int, NV0>{}, std::integral_constant<int, 0>{}, l0);
            scatter(std::integral_constant<uint32_t, 0>{});
            REG2_TICK(2)
            __syncthreads();
            REG2_TICK(3)
            vn_pairs(std::make_integer_sequence<int, NV0 / 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, l0);
            vn_fetch(std::make_integer_sequence<int, NV1>{}, std::integral_constant<int, NV0>{}, l1);
            REG2_TICK(4)
            __syncthreads();
            REG2_TICK(5)
            scatter(std::integral_constant<uint32_t, 1>{});
            REG2_TICK(6)
            __syncthreads();
            REG2_TICK(7)
            vn_pairs(std::make_integer_sequence<int, NV1 / 2>{}, std::integral_constant<int, NV0>{}, std::integral_constant<int, 1>{}, l1);
            REG2_TICK(8)
            __syncthreads();
            REG2_TICK(9)
        }
        else
        {
            scatter(std::integral_constant<uint32_t, 0>{});
            REG2_TICK(2)
            __syncthreads();
            REG2_TICK(3)
            vn_round(std::make_integer_sequence<int, NV0>{}, std::integral_constant<int, 0>{});
            REG2_TICK(4)
            __syncthreads();
            REG2_TICK(5)
            scatter(std::integral_constant<uint32_t, 1>{});
            REG2_TICK(6)
            __syncthreads();
            REG2_TICK(7)
            vn_round(std::make_integer_sequence<int, NV1>{}, std::integral_constant<int, NV0>{});
            REG2_TICK(8)
            __syncthreads();
            REG2_TICK(9)
        }
        ++I;
    }

#ifdef LDPC_AMD_PHASE_TRACE
    if (a.phase_trace && frame >= 2048 && frame < 2048 + 256 && lane == 0)
    {
        uint64_t *o = a.phase_trace + ((frame - 2048) * W + wave) * 16;
        for (int k = 0; k < 10; ++k)
            o[k] = ph[k];
        o[10] = ph_entry - 0, o[11] = __builtin_amdgcn_s_memtime() - ph_entry, o[12] = I;
    }
#endif
    // ---- outputs ----
    // (the thread index afresh: offsets derived from it before the loop would be held — spilled — across the whole decode)
    int tid_out = tid;
    asm volatile("" : "+v"(tid_out));
    const bool ran = a.iterations > 0;
#pragma unroll
    for (int i = 0; i < NV; ++i)
    {
        const Reg2VnBlock vb = load_vn_block(R.vn_blocks, i * W + wave);
        if (lane < vb.count)
        {
            // the totals of the last VN pass are still in LDS: the decision is the entry's sign (ratio form) or out <= 0
            const double t = lds[vb.tot_off + lane];
            const uint32_t bit = RATIO ? (DM_SIGN_WORD(t) >> 31) : (t <= 0);
            hard[R.vn_rank[(i * W + wave) * 64 + lane]] = ran ? bit : 0;
        }
    }
    if (tid == 0)
    {
        vote[2] = 0;
        if (a.iters)
            a.iters[frame] = result;
    }
    __syncthreads(); // hard[] is read back below by other threads of this workgroup
    if (a.hard)
    {
        uint8_t *h = a.hard + frame * nc;
        for (int r = tid_out; r < nc; r += NT)
            h[P.rank_col[r]] = hard[r];
    }
    if constexpr (WANT_LLR)
    {
        if (!ran)
            for (int r = tid_out; r < nc; r += NT)
                out_llr[P.rank_col[r]] = 0.0;
    }
    if (a.bit_errors)
    {
        int err = 0;
        for (int i = tid_out; i < P.n_bitpos; i += NT)
        {
            int est = hard[P.tx_rank[i]];
            int tx = cw ? static_cast<int>(cw[P.bit_pos[i]]) : 0;
            err += est != tx;
        }
        err = wave_sum_i2(err);
        if (lane == 0 && err)
            atomicAdd(&vote[2], static_cast<uint32_t>(err));
        __syncthreads();
        if (tid == 0)
            a.bit_errors[frame] = vote[2];
    }
    return false;
}

// CHAIN: sum-product with early termination — the three forms one after the other, in this workgroup, until one finishes the
// frame (detmath.h: shared reciprocals of the degree-6 check nodes; every output divided separately; the LLR domain.  The
// LDS-resident decoder does the same in three launches over lists; here a frame handed to a later launch would be decoded
// by one workgroup on an otherwise idle chip — the frames that do not converge at the operating point of BASELINE config 4
// leave the box of the ratio form after a few dozen iterations, a handful per batch, and cost 0.3 ms per batch that way —
// while inside the first launch the repeat hides among the other frames.  Which form finishes a frame is the same either way.)
template <bool MINSUM, bool WANT_LLR, int NT, int KC, int MAXD, int NV0, int NV1, bool CHAIN, bool U>
__global__ __launch_bounds__(NT) void decode_reg2_kernel(const DecodeArgs a, const DevReg2Plan R)
{
    static_assert(!(CHAIN && MINSUM), "the chain is the sum-product decoder's");
    extern __shared__ double lds[];
    if (static_cast<uint32_t>(reinterpret_cast<uintptr_t>((double __attribute__((address_space(3))) *)lds)) != 0)
        __builtin_trap(); // the dynamic LDS array does not start at 0: the packed addresses would be wrong
    __builtin_amdgcn_s_setprio(3); // ahead of the noise generator's waves in the SIMD's instruction arbitration (kernels.hip)
    const uint64_t frame = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (CHAIN)
    {
        if (!reg2_frame<false, WANT_LLR, NT, KC, MAXD, NV0, NV1, true, true, U, U>(a, R, frame, wave))
            return;
        __syncthreads(); // the next attempt re-initialises LDS words this one may still be reading
        if (!reg2_frame<false, WANT_LLR, NT, KC, MAXD, NV0, NV1, true, false, U, U>(a, R, frame, wave))
            return;
        __syncthreads();
    }
    reg2_frame<MINSUM, WANT_LLR, NT, KC, MAXD, NV0, NV1, false, false, U, U>(a, R, frame, wave);
}

// U: the regular code's instantiation (no switch over check-node degrees, straight-line variable-node rounds), or the
// generic one
template <int NT, int KC, int MAXD, int NV0, int NV1, bool U>
int launch_reg2(const DecodeArgs &a, const DevReg2Plan &r, bool min_sum, void *stream)
{
    const bool want_llr = a.llr_out != nullptr;
    // the caller's first launch of sum-product with early termination (a list to hand frames back in): the chain kernel,
    // which hands nothing back; its later launches over those lists have nothing to do
    const bool chain = a.redo_list != nullptr;
    if (chain && (min_sum || !a.early_term || a.iterations == 0))
        return hipErrorInvalidValue;
    if (a.redo_count_in)
        return min_sum ? hipErrorInvalidValue : hipSuccess;
    void (*k)(const DecodeArgs, const DevReg2Plan) = nullptr;
    if (min_sum)
        k = want_llr ? decode_reg2_kernel<true, true, NT, KC, MAXD, NV0, NV1, false, U>
                     : decode_reg2_kernel<true, false, NT, KC, MAXD, NV0, NV1, false, U>;
    else if (chain)
        k = want_llr ? decode_reg2_kernel<false, true, NT, KC, MAXD, NV0, NV1, true, U>
                     : decode_reg2_kernel<false, false, NT, KC, MAXD, NV0, NV1, true, U>;
    else
        k = want_llr ? decode_reg2_kernel<false, true, NT, KC, MAXD, NV0, NV1, false, U>
                     : decode_reg2_kernel<false, false, NT, KC, MAXD, NV0, NV1, false, U>;
    const uint32_t lds = r.lds_entries * 8u + 16u;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(a.n_frames)), dim3(NT), lds, static_cast<hipStream_t>(stream), a, r);
    return hipGetLastError();
}

} // namespace

} // namespace ldpc_amd
