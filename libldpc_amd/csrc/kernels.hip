// kernels.hip — hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// Hot path of heat1q/libldpc rebuilt for CDNA4: flooding BP (sum-product / min-sum) with the
// channel + LLR initialisation fused into the same launch.  One workgroup decodes one frame with
// all of its messages resident in LDS; thousands of frames per launch.
//
// Reference semantics restated here (file:line in heat1q/libldpc):
//   decode loop            src/decoding/decoder.cpp:11-78
//   box-plus kernels       src/decoding/decoder.h:7-20
//   syndrome early-term    src/decoding/decoder.h:47-64
//   AWGN channel + LLRs    src/sim/channel.cpp:62-93   (libstdc++ normal_distribution, polar method)
//   BSC channel + LLRs     src/sim/channel.cpp:129-162
//   BEC channel + decoder  src/sim/channel.cpp:199-229, src/decoding/decoder.cpp:91-192
//   bit-error count        src/sim/ldpcsim.cpp:184-188
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "kernels.hpp"

namespace ldpc_amd
{

namespace
{

constexpr int kThreads = kDecodeWaves * kWaveSize;

// ---------------------------------------------------------------------------------------------
// check-node update of one node held by one lane: forward/backward recursion, decoder.cpp:31-44.
// v[j] = v2c of the node's j-th edge (row file order); returns c2v in place.  The reference also
// evaluates F[cw-1] and B[0], which nothing reads; they are skipped.
// ---------------------------------------------------------------------------------------------
template <int D, bool MINSUM>
__device__ __forceinline__ void cn_update(double *m, int stride)
{
    double v[D], F[D], B[D];
#pragma unroll
    for (int j = 0; j < D; ++j)
        v[j] = m[j * stride];
    F[0] = v[0];
    B[D - 1] = v[D - 1];
#pragma unroll
    for (int j = 1; j < D - 1; ++j)
        F[j] = boxplus<MINSUM>(F[j - 1], v[j]);
#pragma unroll
    for (int j = D - 2; j >= 1; --j)
        B[j] = boxplus<MINSUM>(B[j + 1], v[j]);
    m[0] = B[1];
    m[(D - 1) * stride] = F[D - 2];
#pragma unroll
    for (int j = 1; j < D - 1; ++j)
        m[j * stride] = boxplus<MINSUM>(F[j - 1], B[j + 1]);
}

template <bool MINSUM>
__device__ __forceinline__ void cn_block(double *msg, const CnBlock b, int lane)
{
    if (lane >= b.count)
        return;
    double *m = msg + b.off + lane;
    const int s = b.count;
    switch (b.degree) // wave-uniform
    {
    case 2: cn_update<2, MINSUM>(m, s); break;
    case 3: cn_update<3, MINSUM>(m, s); break;
    case 4: cn_update<4, MINSUM>(m, s); break;
    case 5: cn_update<5, MINSUM>(m, s); break;
    case 6: cn_update<6, MINSUM>(m, s); break;
    case 7: cn_update<7, MINSUM>(m, s); break;
    case 8: cn_update<8, MINSUM>(m, s); break;
    default: break;
    }
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// LDS-resident decoder: one workgroup (4 waves) per frame.
//   LDS: msg[nnz] f64 (v2c and c2v share the slot: each edge is rewritten by its own CN lane,
//        then by its own VN lane), llr[nc] f64 (VN rank order), hb[nnz] u8 (hard decision of the
//        edge's VN, read back by the CN lanes for the syndrome).
// ---------------------------------------------------------------------------------------------
template <bool MINSUM, bool WANT_LLR>
__global__ __launch_bounds__(kThreads) void decode_lds_kernel(const DecodeArgs a)
{
    extern __shared__ double lds[];
    const DevPlan &P = a.plan;
    const int nnz = P.nnz, nc = P.nc, nct = P.nct;
    double *msg = lds;
    double *llr = lds + nnz;
    uint8_t *hb = reinterpret_cast<uint8_t *>(llr + nc);
    int *misc = reinterpret_cast<int *>(hb + ((nnz + 15) / 16) * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint64_t frame = blockIdx.x;
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;

    if (tid == 0)
        misc[0] = 0;

    // ---- channel + LLR initialisation (channel.cpp:70-93 / 137-162 / shared.cpp:50-55) ----
    if (a.mode == kModeLlr)
    {
        const double *in = a.llr_in + frame * nc;
        for (int r = tid; r < nc; r += kThreads)
            llr[r] = in[P.rank_col[r]];
    }
    else
    {
        for (int r = tid; r < nc; r += kThreads)
        {
            uint8_t k = P.rank_kind[r];
            if (k == 1)
                llr[r] = 0.0; // punctured = erasure
            else if (k == 2)
                llr[r] = a.shorten_llr;
        }
        if (a.mode == kModeAwgn)
        {
            // normal g of the stream is element (g & 1) of accepted polar pair g >> 1:
            // element 0 = y*mult, element 1 = x*mult (libstdc++ returns y first and saves x)
            const uint64_t g0 = a.normal_base + frame * static_cast<uint64_t>(nct);
            const uint64_t q_lo = g0 >> 1, q_hi = (g0 + nct - 1) >> 1;
            for (uint64_t q = q_lo + tid; q <= q_hi; q += kThreads)
            {
                const uint64_t *pp = a.pairs + 2 * (q - a.pair_base);
                PolarTrial t = polar_trial(pp[0], pp[1]);
                double mult = __builtin_sqrt(-2 * dm_log(t.r2) / t.r2);
                double nrm[2] = {t.y * mult, t.x * mult};
#pragma unroll
                for (int k = 0; k < 2; ++k)
                {
                    uint64_t g = 2 * q + k;
                    if (g < g0 || g >= g0 + nct)
                        continue;
                    int i = static_cast<int>(g - g0);
                    double noise = nrm[k] * a.sigma + 0.0;
                    double xs = cw ? static_cast<double>(1 - 2 * static_cast<int>(cw[P.bit_pos[i]])) : 1.0;
                    double y = noise + xs;
                    llr[P.tx_rank[i]] = 2 * y / a.sigma2;
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
    __syncthreads();

    if (a.llr_in_dump)
    {
        double *o = a.llr_in_dump + frame * nc;
        for (int r = tid; r < nc; r += kThreads)
            o[P.rank_col[r]] = llr[r];
    }

    // ---- v2c initialisation: decoder.cpp:16-19 ----
    const uint16_t *my_vn = P.vn_work + wave * P.vn_work_stride;
    const uint16_t *my_cn = P.cn_work + wave * P.cn_work_stride;
    for (int w = 0; w < P.vn_work_stride; ++w)
    {
        uint16_t bi = my_vn[w];
        if (bi == 0xFFFF)
            break;
        const VnBlock b = P.vn_blocks[bi];
        if (lane < b.count)
        {
            double L = llr[b.first + lane];
            const uint16_t *idx = P.vn_slot + b.idx_off + lane;
            for (int p = 0; p < b.degree; ++p)
                msg[idx[p * b.count]] = L;
        }
    }
    __syncthreads();

    double *out_llr = WANT_LLR ? a.llr_out + frame * nc : nullptr;
    uint32_t I = 0;
    while (I < a.iterations)
    {
        // ---- CN pass: decoder.cpp:25-45 ----
        for (int w = 0; w < P.cn_work_stride; ++w)
        {
            uint16_t bi = my_cn[w];
            if (bi == 0xFFFF)
                break;
            cn_block<MINSUM>(msg, P.cn_blocks[bi], lane);
        }
        __syncthreads();

        // ---- VN pass, APP and hard decision: decoder.cpp:48-64 ----
        for (int w = 0; w < P.vn_work_stride; ++w)
        {
            uint16_t bi = my_vn[w];
            if (bi == 0xFFFF)
                break;
            const VnBlock b = P.vn_blocks[bi];
            if (lane < b.count)
            {
                const int r = b.first + lane;
                const uint16_t *idx = P.vn_slot + b.idx_off + lane;
                double out = llr[r];
                for (int p = 0; p < b.degree; ++p) // sequential sum in column file order
                    out += msg[idx[p * b.count]];
                const uint8_t bit = out <= 0;
                for (int p = 0; p < b.degree; ++p)
                {
                    const int s = idx[p * b.count];
                    msg[s] = out - msg[s];
                    hb[s] = bit;
                }
                if constexpr (WANT_LLR)
                    out_llr[P.rank_col[r]] = out;
            }
        }
        __syncthreads();

        // ---- syndrome early termination: decoder.cpp:66-72, decoder.h:47-64 ----
        if (a.early_term)
        {
            int bad = 0;
            for (int w = 0; w < P.cn_work_stride; ++w)
            {
                uint16_t bi = my_cn[w];
                if (bi == 0xFFFF)
                    break;
                const CnBlock b = P.cn_blocks[bi];
                if (lane < b.count)
                {
                    int par = 0;
                    for (int j = 0; j < b.degree; ++j)
                        par ^= hb[b.off + j * b.count + lane];
                    bad |= par;
                }
            }
            if (!__syncthreads_or(bad))
                break;
        }
        ++I;
    }

    // ---- outputs: iteration count (decoder.cpp:74-77), hard decisions, bit errors (ldpcsim.cpp:184-188) ----
    if (tid == 0 && a.iters)
        a.iters[frame] = I;
    const bool ran = a.iterations > 0;
    auto hard_of_rank = [&](int r) -> int {
        if (!ran)
            return 0; // mCO is still zero-initialised when no iteration ran
        uint16_t s0 = P.rank_slot0[r];
        return s0 != 0xFFFF ? hb[s0] : static_cast<int>(llr[r] <= 0);
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
        else // isolated variable nodes never pass through a VN block with edges
            for (int r = tid; r < nc; r += kThreads)
                if (P.rank_slot0[r] == 0xFFFF)
                    out_llr[P.rank_col[r]] = llr[r];
    }
    if (a.bit_errors)
    {
        int err = 0;
        for (int i = tid; i < nct; i += kThreads)
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
}

} // namespace

int launch_decode_lds(const DecodeArgs &a, bool min_sum, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    const bool want_llr = a.llr_out != nullptr;
    void (*k)(const DecodeArgs) = nullptr;
    if (min_sum)
        k = want_llr ? decode_lds_kernel<true, true> : decode_lds_kernel<true, false>;
    else
        k = want_llr ? decode_lds_kernel<false, true> : decode_lds_kernel<false, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(a.plan.lds_bytes));
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(a.n_frames)), dim3(kThreads), a.plan.lds_bytes,
                       static_cast<hipStream_t>(stream), a);
    return hipGetLastError();
}

} // namespace ldpc_amd
