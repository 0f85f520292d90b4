// device_cn.hpp — check-node update of one node held by one lane.
//
// Forward/backward recursion of the reference (src/decoding/decoder.cpp:31-44) on the D inputs of the node,
// in place: v[j] = v2c of the node's j-th edge (row file order) on entry, c2v on return.  The reference also
// evaluates F[cw-1] and B[0], which nothing reads; they are skipped.
#pragma once

#include <hip/hip_runtime.h>

#include "device_math.hpp"

namespace ldpc_amd
{

// Saturated check node (detmath.h, "Saturated check nodes"): every magnitude at least DM_SAT_MIN and within
// DM_SHARED_LIMIT of the smallest one, mu.  F[j] = inputs 0..j, B[j] = inputs j..D-1, as sums of e^-(|v| - mu).
template <int D>
__device__ __forceinline__ void cn_saturated(double (&v)[D], double mu)
{
    double F[D], B[D];
    uint32_t sv[D], sF[D], sB[D];
#pragma unroll
    for (int j = 0; j < D; ++j)
    {
        F[j] = dm_sat_e(__builtin_fabs(v[j]), mu); // E'_j for now
        sv[j] = DM_SIGN_WORD(v[j]);
    }
    sF[0] = sv[0], sB[D - 1] = sv[D - 1];
    B[D - 1] = F[D - 1];
#pragma unroll
    for (int j = D - 2; j >= 1; --j)
        B[j] = B[j + 1] + F[j], sB[j] = sB[j + 1] ^ sv[j];
#pragma unroll
    for (int j = 1; j < D - 1; ++j)
        F[j] = F[j - 1] + F[j], sF[j] = sF[j - 1] ^ sv[j];
    v[0] = dm_sat_llr(sB[1], mu, B[1]);
    v[D - 1] = dm_sat_llr(sF[D - 2], mu, F[D - 2]);
#pragma unroll
    for (int j = 1; j < D - 1; ++j)
        v[j] = dm_sat_llr(sF[j - 1] ^ sB[j + 1], mu, F[j - 1] + B[j + 1]);
}

template <int D, bool MINSUM>
__device__ __forceinline__ void cn_core(double (&v)[D])
{
    if constexpr (!MINSUM && D > 2) // a degree-2 node only swaps its two inputs: the generic code below
    {
        // sum-product: saturated form when it applies; else the recursion is carried in E = e^-|L| (detmath.h,
        // dm_e_combine / dm_efrac / dm_e_to_llr) while every input is within DM_SHARED_LIMIT; otherwise the direct
        // box-plus below
        double amax = 0.0, mu = __builtin_huge_val();
#pragma unroll
        for (int j = 0; j < D; ++j)
        {
            amax = __builtin_fmax(amax, __builtin_fabs(v[j]));
            mu = __builtin_fmin(mu, __builtin_fabs(v[j]));
        }
        if (mu >= DM_SAT_MIN && ((D >= 5 && D <= 16) || amax - mu <= DM_SHARED_LIMIT))
        {
            // the saturated form; where the inputs' spread has left its window (dm_sat2_applies) the output of the (first)
            // edge that holds the minimum is replaced by its own-base value (detmath.h).  One pass through the form for the
            // whole wave, the replacement computed only by the lanes that need it; the edge is found by comparison and its
            // output chosen by a select: no lane-dependent indexing.
            const bool far = D >= 5 && D <= 16 && dm_sat2_applies(mu, amax);
            uint32_t min_edge = 0; // bit j: edge j is the (first) one that holds the minimum
            double own_mag = 0.0;
            if (far)
            {
                double m2 = __builtin_huge_val();
#pragma unroll
                for (int j = 0; j < D; ++j)
                {
                    const double a = __builtin_fabs(v[j]);
                    const bool is_min = min_edge == 0 && a == mu;
                    min_edge |= static_cast<uint32_t>(is_min) << j;
                    m2 = is_min ? m2 : __builtin_fmin(m2, a);
                }
                double own = 0.0; // sum over the OTHER edges, base m2, index order
#pragma unroll
                for (int j = 0; j < D; ++j)
                    own += (min_edge >> j & 1u) ? 0.0 : dm_sat_e(__builtin_fabs(v[j]), m2);
                own_mag = dm_sat_mag(m2, own);
            }
            cn_saturated<D>(v, mu);
            if (far)
            {
                // (the replaced output's sign is the regular one's: the product of the other inputs' signs either way)
#pragma unroll
                for (int j = 0; j < D; ++j)
                    v[j] = (min_edge >> j & 1u) ? dm_sat_signed(DM_SIGN_WORD(v[j]), own_mag) : v[j];
            }
            return;
        }
        if (amax <= DM_SHARED_LIMIT)
        {
            double ev[D];
            uint32_t sv[D], sF[D], sB[D];
#pragma unroll
            for (int j = 0; j < D; ++j)
            {
                ev[j] = dm_boxplus_exp(__builtin_fabs(v[j]));
                sv[j] = DM_SIGN_WORD(v[j]);
            }
            sF[0] = sv[0], sB[D - 1] = sv[D - 1];
#pragma unroll
            for (int j = 1; j < D - 1; ++j)
                sF[j] = sF[j - 1] ^ sv[j];
#pragma unroll
            for (int j = D - 2; j >= 1; --j)
                sB[j] = sB[j + 1] ^ sv[j];
            if constexpr (D == 3)
            {
                v[0] = dm_e_to_llr(sB[1], dm_e_combine(ev[2], ev[1]));
                v[2] = dm_e_to_llr(sF[1], dm_e_combine(ev[0], ev[1]));
                v[1] = dm_e_to_llr(sF[0] ^ sB[2], dm_e_combine(ev[0], ev[2]));
            }
            else
            {
                // D >= 4: partial results stay undivided fractions (detmath.h, dm_efrac): D divisions, not 3(D-2).
                // F[j] = inputs 0..j (j >= 1), B[j] = inputs j..D-1 (j <= D-2)
                dm_efrac F[D], B[D];
                F[1] = dm_efrac_first(ev[0], ev[1]);
                B[D - 2] = dm_efrac_first(ev[D - 1], ev[D - 2]);
#pragma unroll
                for (int j = 2; j <= D - 2; ++j)
                    F[j] = dm_efrac_step(F[j - 1], ev[j]);
#pragma unroll
                for (int j = D - 3; j >= 1; --j)
                    B[j] = dm_efrac_step(B[j + 1], ev[j]);
                double o[D];
                o[0] = dm_efrac_e(B[1]);
                o[D - 1] = dm_efrac_e(F[D - 2]);
                o[1] = dm_efrac_e(dm_efrac_step(B[2], ev[0]));             // F[0] [+] B[2]
                o[D - 2] = dm_efrac_e(dm_efrac_step(F[D - 3], ev[D - 1])); // F[D-3] [+] B[D-1]
#pragma unroll
                for (int j = 2; j <= D - 3; ++j)
                    o[j] = dm_efrac_e2(F[j - 1], B[j + 1]);
                v[0] = dm_e_to_llr(sB[1], o[0]);
                v[D - 1] = dm_e_to_llr(sF[D - 2], o[D - 1]);
#pragma unroll
                for (int j = 1; j < D - 1; ++j)
                    v[j] = dm_e_to_llr(sF[j - 1] ^ sB[j + 1], o[j]);
            }
            return;
        }
    }
    double F[D], B[D];
    F[0] = v[0];
    B[D - 1] = v[D - 1];
#pragma unroll
    for (int j = 1; j < D - 1; ++j)
        F[j] = boxplus<MINSUM>(F[j - 1], v[j]);
#pragma unroll
    for (int j = D - 2; j >= 1; --j)
        B[j] = boxplus<MINSUM>(B[j + 1], v[j]);
    v[0] = B[1];
    v[D - 1] = F[D - 2];
#pragma unroll
    for (int j = 1; j < D - 1; ++j)
        v[j] = boxplus<MINSUM>(F[j - 1], B[j + 1]);
}

// Likelihood-ratio form of the same recursion (detmath.h, "Likelihood-ratio form"): v[j] = rho(v2c_j) on entry,
// lambda(c2v_j) on return.  Partial results F[j], B[j] are carried as rho; the last box-plus of every output is
// taken directly in lambda form, so a degree-3 node costs three divisions and a node of degree D >= 4 costs D (its
// partial results stay undivided fractions).
// SHARED (early-termination kernels): nodes of degree 3 and 4 take one reciprocal of the product of their denominators
// (detmath.h, "Shared-reciprocal check nodes"); the product's range check joins the frame's escape tracking (escaped).
// SHARED6 (early-termination kernels of codes the LDS-resident decoder does not take): nodes of degree 6 take two
// reciprocals for their six outputs (dm_cn6_shared); that range check is tracked in *escaped6, which the caller lets count
// only once the frame has gone on to the variable-node pass (detmath.h).
template <int D, bool SHARED = false, bool SHARED6 = false>
__device__ __forceinline__ void cn_ratio(double (&v)[D], uint32_t *escaped = nullptr, uint32_t *escaped6 = nullptr)
{
    if constexpr (SHARED6 && D == 6)
    {
        const uint32_t h = dm_cn6_shared(v);
        DM_SHARED_TRACK(*escaped6, h);
    }
    else if constexpr (SHARED && D == 3)
    {
        const uint32_t h = dm_cn3_shared(v);
        DM_SHARED_TRACK(*escaped, h);
    }
    else if constexpr (SHARED && D == 4)
    {
        const uint32_t h = dm_cn4_shared(v);
        DM_SHARED_TRACK(*escaped, h);
    }
    else if constexpr (D == 2)
    {
        const double a = dm_ratio_div(1.0, v[1]), b = dm_ratio_div(1.0, v[0]);
        v[0] = a, v[1] = b;
    }
    else if constexpr (D == 3)
    {
        const double o0 = dm_ratio_lambda(v[2], v[1]); // B[1] = B[2] [+] v[1]
        const double o1 = dm_ratio_lambda(v[0], v[2]); // F[0] [+] B[2]
        const double o2 = dm_ratio_lambda(v[0], v[1]); // F[1] = F[0] [+] v[1]
        v[0] = o0, v[1] = o1, v[2] = o2;
    }
    else if constexpr (D == 4)
    {
        const double nF = DM_FMA(v[0], v[1], 1.0), dF = v[0] + v[1]; // F[1] = nF / dF
        const double nB = DM_FMA(v[3], v[2], 1.0), dB = v[3] + v[2]; // B[2] = nB / dB
        const double o0 = dm_ratio_lambda_frac(nB, dB, v[1]);        // B[1] = B[2] [+] v[1]
        const double o1 = dm_ratio_lambda_frac(nB, dB, v[0]);        // F[0] [+] B[2]
        const double o2 = dm_ratio_lambda_frac(nF, dF, v[3]);        // F[1] [+] B[3]
        const double o3 = dm_ratio_lambda_frac(nF, dF, v[2]);        // F[2] = F[1] [+] v[2]
        v[0] = o0, v[1] = o1, v[2] = o2, v[3] = o3;
    }
    else
    {
        // D >= 5: partial results stay undivided fractions (detmath.h, dm_frac): D divisions, not 3(D-2).
        // F[j] = inputs 0..j, B[j] = inputs j..D-1; a partial over an odd number >= 3 of inputs is rescaled.
        // Order of evaluation: the backward partials first, then one forward sweep that emits every output as soon as
        // its two partials exist and overwrites the input it no longer needs — D-3 stored fractions and one running
        // one instead of 2(D-3) fractions plus D pending outputs (the register-resident kernels have no registers to
        // spare; the fences keep the scheduler from interleaving the steps again).  Same operations, same values.
        dm_frac B[D];
        B[D - 2] = dm_frac_first(v[D - 1], v[D - 2]);
#pragma unroll
        for (int j = D - 3; j >= 2; --j)
        {
            B[j] = dm_frac_step(B[j + 1], v[j]);
            if ((D - j) % 2 == 1)
                B[j] = dm_frac_norm(B[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const double o0 = dm_ratio_lambda_frac(B[2].n, B[2].d, v[1]); // B[1] = B[2] [+] v[1]
            const double o1 = dm_ratio_lambda_frac(B[2].n, B[2].d, v[0]); // F[0] [+] B[2]
            dm_frac F = dm_frac_first(v[0], v[1]);                        // F[1]
            v[0] = o0, v[1] = o1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 2; j <= D - 3; ++j)
            {
                const double o = dm_frac_lambda2(F, B[j + 1]); // F[j-1] [+] B[j+1]
                F = dm_frac_step(F, v[j]);                     // F[j]
                if ((j + 1) % 2 == 1)
                    F = dm_frac_norm(F);
                v[j] = o;
                __builtin_amdgcn_sched_barrier(0);
            }
            const double oa = dm_ratio_lambda_frac(F.n, F.d, v[D - 1]); // F[D-3] [+] B[D-1]
            const double ob = dm_ratio_lambda_frac(F.n, F.d, v[D - 2]); // F[D-2] = F[D-3] [+] v[D-2]
            v[D - 2] = oa, v[D - 1] = ob;
        }
    }
}

} // namespace ldpc_amd
