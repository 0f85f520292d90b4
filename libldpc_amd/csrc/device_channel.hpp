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
        // ones at the head of every frame
        for (int r8 = 8 * tid; r8 < nc; r8 += 8 * kThreads)
        {
            const uint64_t kinds = *reinterpret_cast<const uint64_t *>(P.rank_kind + r8);
            if (kinds == 0)
                continue; // (eight transmitted bits: the channel writes them)
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
        }
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
            constexpr int kGroup = 2;
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
        }
    }
}

} // namespace ldpc_amd
