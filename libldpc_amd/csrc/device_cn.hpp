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

template <int D, bool MINSUM>
__device__ __forceinline__ void cn_core(double (&v)[D])
{
    if constexpr (!MINSUM && D > 2) // a degree-2 node only swaps its two inputs: the generic code below
    {
        // sum-product: the recursion is carried in E = e^-|L| (detmath.h, dm_e_combine / dm_e_to_llr) while
        // every input is within DM_SHARED_LIMIT; otherwise the direct box-plus below
        double amax = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j)
            amax = __builtin_fmax(amax, __builtin_fabs(v[j]));
        if (amax <= DM_SHARED_LIMIT)
        {
            double ev[D], eF[D], eB[D];
            uint32_t sv[D], sF[D], sB[D];
#pragma unroll
            for (int j = 0; j < D; ++j)
            {
                ev[j] = dm_boxplus_exp(__builtin_fabs(v[j]));
                sv[j] = DM_SIGN_WORD(v[j]);
            }
            eF[0] = ev[0], sF[0] = sv[0];
            eB[D - 1] = ev[D - 1], sB[D - 1] = sv[D - 1];
#pragma unroll
            for (int j = 1; j < D - 1; ++j)
            {
                eF[j] = dm_e_combine(eF[j - 1], ev[j]);
                sF[j] = sF[j - 1] ^ sv[j];
            }
#pragma unroll
            for (int j = D - 2; j >= 1; --j)
            {
                eB[j] = dm_e_combine(eB[j + 1], ev[j]);
                sB[j] = sB[j + 1] ^ sv[j];
            }
            v[0] = dm_e_to_llr(sB[1], eB[1]);
            v[D - 1] = dm_e_to_llr(sF[D - 2], eF[D - 2]);
#pragma unroll
            for (int j = 1; j < D - 1; ++j)
                v[j] = dm_e_to_llr(sF[j - 1] ^ sB[j + 1], dm_e_combine(eF[j - 1], eB[j + 1]));
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

} // namespace ldpc_amd
