// device_channel.hpp — channel + LLR initialisation fused into the decoder launches.
//
// Restates channel_awgn / channel_bsc ::simulate + ::calculate_llrs (src/sim/channel.cpp:62-93, 129-162) and
// the scatter of the C-ABI decode() (src/shared.cpp:50-55) for one frame, executed by the NT threads of the
// frame's workgroup.  `llr` is the frame's input-LLR array in VN-rank order (LDS or device memory).
#pragma once

#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "kernels.hpp"

namespace ldpc_amd
{

template <int NT>
__device__ __forceinline__ void channel_init(const DecodeArgs &a, uint64_t frame, double *llr, int tid)
{
    const DevPlan &P = a.plan;
    const int nc = P.nc, nct = P.nct;
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;
    constexpr int kThreads = NT;
    // ---- channel + LLR initialisation (channel.cpp:70-93 / 137-162 / shared.cpp:50-55) ----
    if (a.mode == kModeLlr)
    {
        const double *in = a.llr_in + frame * nc;
        for (int r = tid; r < nc; r += kThreads)
            llr[r] = in[P.rank_col[r]];
    }
    else
    {
        // the kinds of eight ranks per load (the table is padded to a multiple of eight, plan.cpp): one round trip for codes
        // of up to 8 x the workgroup's threads variable nodes, where a rank per trip of the loop was nc / threads dependent
        // ones at the head of every frame.  The first load goes out here and is looked at after the channel's own loads
        // have gone out: one more round trip that overlaps.
        const uint64_t kinds_first = 8 * tid < nc ? *reinterpret_cast<const uint64_t *>(P.rank_kind + 8 * tid) : 0;
        auto apply_kinds = [&](int r8, uint64_t kinds) {
            if (kinds == 0)
                return; // (eight transmitted bits: the channel writes them)
#pragma unroll
            for (int e = 0; e < 8; ++e)
            {
                const uint32_t k = static_cast<uint32_t>(kinds >> (8 * e)) & 0xFFu;
                const int r = r8 + e;
                if (r >= nc)
                    break;
                if (k == 1)
                    llr[r] = 0.0; // punctured = erasure
                else if (k == 2)
                    llr[r] = a.shorten_llr;
                else if (k == 3)
                    llr[r] = 0.0; // never written by the channel: keeps the decoder's initial zero
            }
        };
        auto finish_kinds = [&] {
            apply_kinds(8 * tid, kinds_first);
            for (int r8 = 8 * (tid + kThreads); r8 < nc; r8 += 8 * kThreads)
                apply_kinds(r8, *reinterpret_cast<const uint64_t *>(P.rank_kind + r8));
        };
        if (a.mode == kModeAwgn)
        {
            // normal g of the stream is element (g & 1) of accepted polar pair g >> 1:
            // element 0 = y*mult, element 1 = x*mult (libstdc++ returns y first and saves x; channel.cpp:62-68)
            const uint64_t g0 = a.normal_base + frame * static_cast<uint64_t>(nct);
            const uint64_t q_lo = g0 >> 1, q_hi = (g0 + nct - 1) >> 1;
            // The noise generator leaves the accepted pairs in one slab per generator chunk (rng_kernels.hip): find the slab
            // of the frame's first pair — a first guess from the expected pairs per slab, corrected on the table of
            // cumulative counts (wave-uniform: scalar loads); a frame's pairs then lie in that slab or the ones after it.
            const auto cum = uniform_table(a.slab_cum);
            const uint64_t rel_lo = q_lo - a.pair_origin, rel_hi = q_hi - a.pair_origin;
            uint32_t j0 = static_cast<uint32_t>(static_cast<float>(rel_lo) * a.slab_pairs_inv);
            j0 = j0 < a.n_slabs ? j0 : a.n_slabs - 1;
            // four consecutive table entries around the guess, fetched together (the table is followed by four entries of
            // 2^64-1): ONE round trip to memory covers the guess being off by one slab either way — a decode workgroup that
            // owns its CU (the n=8192 code) has no other frame to hide a chain of dependent loads behind
            const uint32_t jb = j0 > 0 ? j0 - 1 : 0;
            const uint64_t c0 = cum[jb], c1 = cum[jb + 1], c2 = cum[jb + 2], c3 = cum[jb + 3];
            const bool window = c0 <= rel_lo && rel_hi < c3;
            if (!window) // (a guess off by more than one slab: walk the table)
            {
                while (j0 > 0 && rel_lo < cum[j0])
                    --j0;
                while (j0 + 1 < a.n_slabs && rel_lo >= cum[j0 + 1])
                    ++j0;
            }
            // The pairs of a frame in groups of kGroup per thread: the loads of a group — the pair words from the slabs and
            // the rank of each normal's bit, which depends on the pair's index only — all go out before anything waits for
            // one of them.  (One pair per trip of a plain loop: a trip's first use sits in front of the next trip's load, two
            // or three round trips to memory one after the other at the head of every frame — and a frame holds its place
            // on the CU for as long as its prologue takes, whatever the other frames there do meanwhile.)
            constexpr int kGroup = 3;
            bool kinds_done = false;
            for (uint64_t q0 = q_lo + tid; q0 <= q_hi; q0 += static_cast<uint64_t>(kGroup) * kThreads)
            {
                ulonglong2 pp[kGroup];
                uint32_t rk[kGroup][2];
                int xb[kGroup][2];
#pragma unroll
                for (int u = 0; u < kGroup; ++u)
                {
                    const uint64_t q = q0 + static_cast<uint64_t>(u) * kThreads;
                    pp[u].x = pp[u].y = 0;
                    rk[u][0] = rk[u][1] = 0, xb[u][0] = xb[u][1] = 0;
                    if (q > q_hi)
                        continue;
                    const uint64_t rel = q - a.pair_origin;
                    uint32_t j;
                    uint64_t base;
                    if (window)
                    {
                        const uint32_t k = (rel >= c1) + (rel >= c2);
                        j = jb + k;
                        base = k == 0 ? c0 : (k == 1 ? c1 : c2);
                    }
                    else
                    {
                        j = j0;
                        while (j + 1 < a.n_slabs && rel >= a.slab_cum[j + 1])
                            ++j;
                        base = a.slab_cum[j];
                    }
                    // streamed once: non-temporal, so that the 8 KB of a frame do not push the slot tables — which every
                    // frame that starts on this CU reads — out of the CU's 32 KB vector cache
                    const uint64_t *pq = a.pairs + static_cast<uint64_t>(j) * a.slab_words + 2 * (rel - base);
                    pp[u].x = __builtin_nontemporal_load(pq), pp[u].y = __builtin_nontemporal_load(pq + 1);
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                    {
                        const uint64_t g = 2 * q + k;
                        if (g < g0 || g >= g0 + nct)
                            continue;
                        const int i = static_cast<int>(g - g0);
                        rk[u][k] = P.tx_rank[i];
                        if (cw)
                            xb[u][k] = static_cast<int>(cw[P.bit_pos[i]]);
                    }
                }
                if (!kinds_done)
                    finish_kinds(), kinds_done = true; // (ranks the channel does not write: disjoint from the ones written below)
#pragma unroll
                for (int u = 0; u < kGroup; ++u)
                {
                    const uint64_t q = q0 + static_cast<uint64_t>(u) * kThreads;
                    if (q > q_hi)
                        continue;
                    const double nrm[2] = {dm_from_bits(pp[u].x), dm_from_bits(pp[u].y)};
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                    {
                        const uint64_t g = 2 * q + k;
                        if (g < g0 || g >= g0 + nct)
                            continue;
                        const double noise = nrm[k] * a.sigma + 0.0;
                        const double xs = cw ? static_cast<double>(1 - 2 * xb[u][k]) : 1.0;
                        const double y = noise + xs;
                        llr[rk[u][k]] = 2 * y / a.sigma2;
                    }
                }
            }
            if (!kinds_done)
                finish_kinds(); // (a thread without a pair of its own: small codes)
        }
        else // kModeBsc
        {
            const uint64_t *raw = a.raw + frame * static_cast<uint64_t>(nct);
            for (int i = tid; i < nct; i += kThreads)
            {
                int flip = canonical(raw[i]) < a.eps;
                int xb = cw ? static_cast<int>(cw[P.bit_pos[i]]) : 0;
                int y = xb ^ flip;
                llr[P.tx_rank[i]] = a.delta * static_cast<double>(1 - 2 * y);
            }
            finish_kinds();
        }
    }
}

// The same channel, a LANE at a time (LDS-resident decoder with the input LLRs in registers): L[w] = the input LLR of this
// lane's variable node in block w of its wave's work list, straight from where it comes from — no staging of the frame's LLRs
// in LDS, no barrier between the channel and the first variable-node work, and every load of a lane (up to VNB normals)
// in flight at once.  pk = the wave's rows of the plan's packed table at this lane (plan.hpp, vn_packed): rows 16.. = the
// index of the node's bit among the transmitted ones or a kVnSrc* code, rows 24.. = its column.  Same arithmetic, same
// values as channel_init.  (The columns are fetched only where they are needed — given LLRs, a transmitted codeword — and
// the indices only where they are: every word held across the loads is a register the headline kernel does not have.)
template <int VNB>
__device__ __forceinline__ void channel_lanes(const DecodeArgs &a, uint64_t frame, const uint32_t *pk, double (&L)[VNB + 1])
{
    const DevPlan &P = a.plan;
    const int nc = P.nc, nct = P.nct;
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;
#pragma unroll
    for (int w = 0; w < VNB; ++w)
        L[w] = 0.0;
    if (a.mode == kModeLlr)
    {
        const double *in = a.llr_in + frame * nc;
#pragma unroll
        for (int w = 0; w < VNB; ++w)
        {
            const uint32_t col = pk[(24 + w) * kWaveSize];
            if (col != kVnSrcZero)
                L[w] = in[col];
        }
        return;
    }
    uint32_t src[VNB];
#pragma unroll
    for (int w = 0; w < VNB; ++w)
        src[w] = pk[(16 + w) * kWaveSize];
    int xb[VNB];
#pragma unroll
    for (int w = 0; w < VNB; ++w)
    {
        xb[w] = 0;
        if (cw && src[w] < kVnSrcShortened)
            xb[w] = static_cast<int>(cw[pk[(24 + w) * kWaveSize]]); // (bit_pos[i] is the node's column)
    }
    if (a.mode == kModeAwgn)
    {
        // normal g of the stream is element (g & 1) of accepted polar pair g >> 1 (channel_init): the slab of the frame's
        // first pair from a first guess corrected on the table of cumulative counts, four entries in one round trip
        const uint64_t g0 = a.normal_base + frame * static_cast<uint64_t>(nct);
        const uint64_t q_lo = g0 >> 1, q_hi = (g0 + nct - 1) >> 1;
        const auto cum = uniform_table(a.slab_cum);
        const uint64_t rel_lo = q_lo - a.pair_origin, rel_hi = q_hi - a.pair_origin;
        uint32_t j0 = static_cast<uint32_t>(static_cast<float>(rel_lo) * a.slab_pairs_inv);
        j0 = j0 < a.n_slabs ? j0 : a.n_slabs - 1;
        const uint32_t jb = j0 > 0 ? j0 - 1 : 0;
        const uint64_t c0 = cum[jb], c1 = cum[jb + 1], c2 = cum[jb + 2], c3 = cum[jb + 3];
        const bool window = c0 <= rel_lo && rel_hi < c3;
        if (!window) // (a guess off by more than one slab: walk the table)
        {
            while (j0 > 0 && rel_lo < cum[j0])
                --j0;
            while (j0 + 1 < a.n_slabs && rel_lo >= cum[j0 + 1])
                ++j0;
        }
        uint64_t nb[VNB];
        if (window) // (wave-uniform) straight-line: VNB addresses, VNB loads in flight; a lane without a transmitted bit in
        {           // block w reads the slab's first word and ignores it
#pragma unroll
            for (int w = 0; w < VNB; ++w)
            {
                const bool tx = src[w] < kVnSrcShortened;
                const uint64_t g = g0 + (tx ? src[w] : 0u);
                const uint64_t rel = (g >> 1) - a.pair_origin;
                const uint32_t k = (rel >= c1) + (rel >= c2);
                const uint64_t base = k == 0 ? c0 : (k == 1 ? c1 : c2);
                nb[w] = __builtin_nontemporal_load(a.pairs + static_cast<uint64_t>(jb + k) * a.slab_words + 2 * (rel - base) + (g & 1));
            }
        }
        else
        {
#pragma unroll 1
            for (int w = 0; w < VNB; ++w)
            {
                nb[w] = 0;
                if (src[w] >= kVnSrcShortened)
                    continue;
                const uint64_t g = g0 + src[w];
                const uint64_t rel = (g >> 1) - a.pair_origin;
                uint32_t j = j0;
                while (j + 1 < a.n_slabs && rel >= a.slab_cum[j + 1])
                    ++j;
                nb[w] = __builtin_nontemporal_load(a.pairs + static_cast<uint64_t>(j) * a.slab_words + 2 * (rel - a.slab_cum[j]) + (g & 1));
            }
        }
#pragma unroll
        for (int w = 0; w < VNB; ++w)
        {
            if (src[w] == kVnSrcShortened)
                L[w] = a.shorten_llr;
            else if (src[w] != kVnSrcZero)
            {
                const double noise = dm_from_bits(nb[w]) * a.sigma + 0.0;
                const double xs = cw ? static_cast<double>(1 - 2 * xb[w]) : 1.0;
                const double y = noise + xs;
                L[w] = 2 * y / a.sigma2;
            }
        }
    }
    else // kModeBsc
    {
        const uint64_t *raw = a.raw + frame * static_cast<uint64_t>(nct);
        uint64_t rw[VNB];
#pragma unroll
        for (int w = 0; w < VNB; ++w)
            rw[w] = raw[src[w] < kVnSrcShortened ? src[w] : 0u]; // (a lane without a transmitted bit reads word 0 and ignores it)
#pragma unroll
        for (int w = 0; w < VNB; ++w)
        {
            if (src[w] == kVnSrcShortened)
                L[w] = a.shorten_llr;
            else if (src[w] != kVnSrcZero)
            {
                const int flip = canonical(rw[w]) < a.eps;
                const int y = xb[w] ^ flip;
                L[w] = a.delta * static_cast<double>(1 - 2 * y);
            }
        }
    }
}

} // namespace ldpc_amd
