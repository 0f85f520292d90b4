// kernels_fast.hip — the opt-in NON-PARITY fast mode: sum-product decoding with binary32
// messages.  Results are NOT those of the reference: the mode exists for throughput studies and is never used unless
// the caller asks for it (ldpc_hip_set_fast_mode); every parity test runs the binary64 kernels.
//
// Same schedule as the reference (flooding, CN pass then VN pass, syndrome early termination, decoder.cpp:11-78) and
// the same data layout as the LDS-resident kernel (plan.hpp: lane = node, blocks of <= 64 equal-degree nodes), but
//   * messages are floats in LDS (4 bytes per edge): v2c as rho = 2^L2 carrying the node's hard decision in its sign
//     bit, c2v as lambda = 2^-L2, with L2 the LLR in log2 units clipped to +-kClip (a usual fixed-range decoder);
//   * the check node works on ratios, (a + b) / (1 + a b) with the hardware reciprocal (v_rcp_f32), in the
//     reference's forward/backward order; the variable node sums log2 values (v_log_f32 / v_exp_f32), so no product
//     of many ratios is ever formed and no frame needs a second pass;
//   * binary32 vector instructions issue at twice the binary64 rate and a transcendental is one instruction.
// What it costs in error rate is measured by tools/fast_mode_report.py (profiles/): FER / BER against the binary64
// path over >= 10^6 frames.
#include <hip/hip_runtime.h>

#include "device_channel.hpp"
#include "device_math.hpp"
#include "kernels.hpp"

namespace ldpc_amd
{

namespace
{
constexpr int kFastThreads = 256, kFastWaves = 4;
constexpr int kFastBlocks = 8; // VN blocks per wave the kernel takes (plan.vn_work_stride)
constexpr float kClip = 40.0f;                 // |L2| <= 40: |LLR| <= 27.7
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ float f_abs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ uint32_t f_bits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float f_from(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float f_log2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32
__device__ __forceinline__ float f_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32

// lambda(a [+] b) and rho(a [+] b) for ratios a, b in [2^-40, 2^40]
__device__ __forceinline__ float f_lambda(float a, float b) { return (a + b) * f_rcp(__builtin_fmaf(a, b, 1.0f)); }
__device__ __forceinline__ float f_rho(float a, float b) { return __builtin_fmaf(a, b, 1.0f) * f_rcp(a + b); }

// check node of degree D on the lane's D slots (stride = nodes in the block): rho in (sign = hard decision of the
// sending variable node), lambda out; returns the parity of the decisions (syndrome bit of this check)
template <int D>
__device__ __forceinline__ uint32_t cn_fast(float *m, int stride)
{
    float v[D];
    uint32_t par = 0;
#pragma unroll
    for (int j = 0; j < D; ++j)
    {
        const float x = m[j * stride];
        par ^= f_bits(x);
        v[j] = f_abs(x);
    }
    float o[D];
    if constexpr (D == 2)
    {
        o[0] = f_rcp(v[1]), o[1] = f_rcp(v[0]);
    }
    else
    {
        float F[D], B[D]; // partial results as rho, the reference's recursion order (decoder.cpp:31-44)
        F[0] = v[0], B[D - 1] = v[D - 1];
#pragma unroll
        for (int j = 1; j <= D - 3; ++j)
            F[j] = f_rho(F[j - 1], v[j]);
#pragma unroll
        for (int j = D - 2; j >= 2; --j)
            B[j] = f_rho(B[j + 1], v[j]);
        o[0] = f_lambda(D > 3 ? B[2] : v[2], v[1]);
        o[D - 1] = f_lambda(D > 3 ? F[D - 3] : v[0], v[D - 2]);
#pragma unroll
        for (int j = 1; j < D - 1; ++j)
            o[j] = f_lambda(F[j - 1], B[j + 1]);
    }
#pragma unroll
    for (int j = 0; j < D; ++j)
        m[j * stride] = o[j];
    return par >> 31;
}

__device__ __forceinline__ uint32_t cn_fast_block(float *msg, const CnBlock b, int lane)
{
    if (lane >= b.count)
        return 0;
    float *m = msg + b.off + lane;
    switch (b.degree) // wave-uniform
    {
    case 2: return cn_fast<2>(m, b.count);
    case 3: return cn_fast<3>(m, b.count);
    case 4: return cn_fast<4>(m, b.count);
    case 5: return cn_fast<5>(m, b.count);
    case 6: return cn_fast<6>(m, b.count);
    case 7: return cn_fast<7>(m, b.count);
    case 8: return cn_fast<8>(m, b.count);
    default: return 0;
    }
}

// variable node of degree DV (decoder.cpp:48-64) in log2 units on the slots sl[]: every message read once, all loads in
// flight together; returns the total, writes rho(total - c2v_p) with the decision in the sign bit
template <int DV>
__device__ __forceinline__ float vn_fast(float *msg, const uint32_t (&sl)[DV], float l2ch)
{
    float c[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        c[p] = msg[sl[p]];
    float tot = l2ch;
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        c[p] = f_log2(c[p]); // c2v is lambda = 2^-L2: -log2 is the message in log2 units
        tot -= c[p];         // sequential sum in column file order
    }
    const uint32_t sign = tot <= 0.0f ? 0x80000000u : 0u;
#pragma unroll
    for (int p = 0; p < DV; ++p)
    {
        const float e = __builtin_fminf(__builtin_fmaxf(tot + c[p], -kClip), kClip); // total - c2v_p, clipped
        msg[sl[p]] = f_from(f_bits(f_exp2(e)) | sign);
    }
    return tot;
}

// the same with the slot indices taken from a table (idx[p * count]) or from packed 16-bit halves held in registers
template <int DV>
__device__ __forceinline__ float vn_fast_table(float *msg, const uint32_t *idx, int count, float l2ch)
{
    uint32_t sl[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        sl[p] = idx[p * count];
    return vn_fast<DV>(msg, sl, l2ch);
}

template <int DV>
__device__ __forceinline__ float vn_fast_packed(float *msg, const uint32_t *packed, float l2ch)
{
    uint32_t pk[(DV + 1) / 2];
#pragma unroll
    for (int i = 0; i < (DV + 1) / 2; ++i)
    {
        pk[i] = packed[i];
        asm volatile("" : "+v"(pk[i])); // unpack here, every iteration: the packed words are what stays live
    }
    uint32_t sl[DV];
#pragma unroll
    for (int p = 0; p < DV; ++p)
        sl[p] = (p & 1) ? pk[p >> 1] >> 16 : pk[p >> 1] & 0xFFFFu;
    return vn_fast<DV>(msg, sl, l2ch);
}

__device__ __forceinline__ float vn_fast_any(float *msg, const uint32_t *idx, int count, int degree, float l2ch)
{
    switch (degree) // wave-uniform
    {
#define LDPC_VN(D) case D: return vn_fast_table<D>(msg, idx, count, l2ch);
        LDPC_VN(1) LDPC_VN(2) LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7) LDPC_VN(8)
        LDPC_VN(9) LDPC_VN(10) LDPC_VN(11) LDPC_VN(12) LDPC_VN(13) LDPC_VN(14) LDPC_VN(15) LDPC_VN(16)
#undef LDPC_VN
    default:
    {
        float tot = l2ch;
        for (int p = 0; p < degree; ++p)
            tot -= f_log2(msg[idx[p * count]]);
        const uint32_t sign = tot <= 0.0f ? 0x80000000u : 0u;
        for (int p = 0; p < degree; ++p)
        {
            const uint32_t s = idx[p * count];
            const float e = __builtin_fminf(__builtin_fmaxf(tot + f_log2(msg[s]), -kClip), kClip);
            msg[s] = f_from(f_bits(f_exp2(e)) | sign);
        }
        return tot;
    }
    }
}

__device__ __forceinline__ float vn_fast_wide(float *msg, const uint32_t (&packed)[8], int degree, float l2ch)
{
    switch (degree) // wave-uniform, 3..16
    {
#define LDPC_VN(D) case D: return vn_fast_packed<D>(msg, packed, l2ch);
        LDPC_VN(3) LDPC_VN(4) LDPC_VN(5) LDPC_VN(6) LDPC_VN(7) LDPC_VN(8) LDPC_VN(9) LDPC_VN(10)
        LDPC_VN(11) LDPC_VN(12) LDPC_VN(13) LDPC_VN(14) LDPC_VN(15) LDPC_VN(16)
#undef LDPC_VN
    default: return 0.0f;
    }
}

__device__ __forceinline__ int wave_sum_fast(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

template <bool WANT_LLR>
__global__ __launch_bounds__(kFastThreads) void decode_fast_kernel(const DecodeArgs a)
{
    extern __shared__ double lds_d[];
    __shared__ int misc[4];
    __shared__ int votes[2][kFastWaves];
    const DevPlan &P = a.plan;
    const int nnz = P.nnz, nc = P.nc;
    const uint64_t frame = blockIdx.x;
    // LDS: float msg[nnz] (the channel's double LLRs pass through the same bytes first: 4 nnz >= 8 nc), float l2[nc]
    float *msg = reinterpret_cast<float *>(lds_d);
    double *llr = lds_d;
    float *l2 = reinterpret_cast<float *>(lds_d) + ((nnz + 3) & ~3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;
    if (tid == 0)
        misc[0] = 0;

    channel_init<kFastThreads>(a, frame, llr, tid); // binary64 channel + LLR initialisation, as everywhere
    __syncthreads();
    if (a.llr_in_dump)
    {
        double *o = a.llr_in_dump + frame * nc;
        for (int r = tid; r < nc; r += kFastThreads)
            o[P.rank_col[r]] = llr[r];
    }
    float mine[(8192 + kFastThreads - 1) / kFastThreads]; // this thread's channel values in log2 units (nc <= 8192)
#pragma unroll
    for (int i = 0; i < static_cast<int>(sizeof mine / sizeof mine[0]); ++i)
    {
        const int r = tid + i * kFastThreads;
        mine[i] = r < nc ? __builtin_fminf(__builtin_fmaxf(static_cast<float>(llr[r]) * kLog2e, -kClip), kClip) : 0.0f;
    }
    __syncthreads(); // everyone has read the doubles: the bytes become the message array
#pragma unroll
    for (int i = 0; i < static_cast<int>(sizeof mine / sizeof mine[0]); ++i)
    {
        const int r = tid + i * kFastThreads;
        if (r < nc)
            l2[r] = mine[i];
    }
    __syncthreads();

    const auto my_vdesc = uniform_table(P.vn_work_desc + wave * (P.vn_work_stride + 1) * 4);
    const auto my_cdesc = uniform_table(reinterpret_cast<const uint32_t *>(P.cn_work_desc + wave * P.cn_desc_stride));
    auto vn_desc = [&](int w) {
        const uint32_t d0 = my_vdesc[4 * w], d1 = my_vdesc[4 * w + 1], d2 = my_vdesc[4 * w + 2];
        return VnBlock{d0, d1, static_cast<uint16_t>(d2 & 0xFFFFu), static_cast<uint16_t>(d2 >> 16)};
    };
    auto cn_desc = [&](int w) {
        const uint32_t d0 = my_cdesc[2 * w], d1 = my_cdesc[2 * w + 1];
        return CnBlock{d0, static_cast<uint16_t>(d1 & 0xFFFFu), static_cast<uint16_t>(d1 >> 16)};
    };
    // Per lane, for the (at most kFastBlocks) VN blocks of this wave: the node's channel value and — as in the binary64
    // kernel — the slot indices of nodes of degree <= 2 (two u16 in a word) and of the wave's first block when it has
    // up to 16 edges per node (eight words); other blocks go through the slot table.
    float my_l2[kFastBlocks];
    uint32_t my_idx[kFastBlocks];
    uint32_t wide_idx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        wide_idx[i] = 0;
#pragma unroll
    for (int w = 0; w < kFastBlocks; ++w)
    {
        my_l2[w] = 0.0f, my_idx[w] = 0;
        if (w < P.vn_work_stride)
        {
            const VnBlock b = vn_desc(w);
            if (b.count && lane < b.count)
            {
                my_l2[w] = l2[b.first + lane];
                const uint32_t *idx = P.vn_slot + b.idx_off + lane;
                if (b.degree >= 1 && b.degree <= 2)
                    my_idx[w] = idx[0] | (idx[(b.degree - 1) * b.count] << 16);
                else if (w == 0 && b.degree <= 16)
                {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (q < b.degree)
                            wide_idx[q >> 1] |= idx[q * b.count] << (16 * (q & 1));
                }
            }
        }
    }
    // v2c initialisation (decoder.cpp:16-19): rho(L_ch) on every edge, decision bit clear
    for (int w = 0; w < P.vn_work_stride; ++w)
    {
        const VnBlock b = vn_desc(w);
        if (b.count == 0)
            break;
        if (lane < b.count)
        {
            const float v0 = f_exp2(l2[b.first + lane]);
            const uint32_t *idx = P.vn_slot + b.idx_off + lane;
            for (int p = 0; p < b.degree; ++p)
                msg[idx[p * b.count]] = v0;
        }
    }
    __syncthreads();

    double *out_llr = WANT_LLR ? a.llr_out + frame * nc : nullptr;
    uint8_t *hb = a.ws_hb + frame * nc; // hard decisions of the last VN pass, VN-rank order
    uint32_t I = 0;
    for (;;)
    {
        // CN pass of iteration I; its inputs' sign bits are the decisions of VN pass I-1: their parity is the syndrome
        uint32_t bad = 0;
        for (int w = 0; w < P.cn_work_stride; ++w)
        {
            const CnBlock b = cn_desc(w);
            if (b.count == 0)
                break;
            bad |= cn_fast_block(msg, b, lane);
        }
        const int ph = I & 1;
        const int wave_vote = __ballot(bad != 0) != 0; // (by all lanes, outside the branch)
        if (lane == 0)
            votes[ph][wave] = wave_vote;
        __syncthreads();
        int any = 0;
#pragma unroll
        for (int w = 0; w < kFastWaves; ++w)
            any |= votes[ph][w];
        if (I > 0 && a.early_term && !any) // decoder.cpp:66-72 after VN pass I-1
        {
            --I;
            break;
        }
        if (I == a.iterations)
            break;
        // VN pass (decoder.cpp:48-64) in log2 units
#pragma unroll
        for (int w = 0; w < kFastBlocks; ++w)
        {
            if (w >= P.vn_work_stride)
                break;
            const VnBlock b = vn_desc(w);
            if (b.count == 0)
                break;
            if (lane < b.count)
            {
                float tot;
                if (b.degree == 0)
                    tot = my_l2[w];
                else if (b.degree <= 2)
                {
                    uint32_t pk = my_idx[w];
                    asm volatile("" : "+v"(pk));
                    if (b.degree == 1)
                    {
                        const uint32_t sl[1] = {pk & 0xFFFFu};
                        tot = vn_fast<1>(msg, sl, my_l2[w]);
                    }
                    else
                    {
                        const uint32_t sl[2] = {pk & 0xFFFFu, pk >> 16};
                        tot = vn_fast<2>(msg, sl, my_l2[w]);
                    }
                }
                else if (w == 0 && b.degree <= 16)
                    tot = vn_fast_wide(msg, wide_idx, b.degree, my_l2[w]);
                else
                    tot = vn_fast_any(msg, P.vn_slot + b.idx_off + lane, b.count, b.degree, my_l2[w]);
                hb[b.first + lane] = static_cast<uint8_t>(tot <= 0.0f); // the node's hard decision (decoder.cpp:58)
                if constexpr (WANT_LLR)
                    out_llr[P.rank_col[b.first + lane]] = static_cast<double>(tot) * 0.6931471805599453;
            }
        }
        __syncthreads();
        ++I;
    }

    if (tid == 0 && a.iters)
        a.iters[frame] = I;
    const bool ran = a.iterations > 0;
    __syncthreads(); // hb[] (written by the last VN pass, or cleared below) is read by other threads from here on
    if (!ran)
    {
        for (int r = tid; r < nc; r += kFastThreads)
            hb[r] = 0; // mCO is still zero-initialised when no iteration ran
        __syncthreads();
    }
    if (a.hard)
    {
        uint8_t *h = a.hard + frame * nc;
        for (int r = tid; r < nc; r += kFastThreads)
            h[P.rank_col[r]] = hb[r];
    }
    if constexpr (WANT_LLR)
    {
        if (!ran)
            for (int r = tid; r < nc; r += kFastThreads)
                out_llr[P.rank_col[r]] = 0.0;
    }
    if (a.bit_errors)
    {
        int err = 0;
        for (int i = tid; i < P.n_bitpos; i += kFastThreads)
        {
            const int est = hb[P.tx_rank[i]];
            const int tx = cw ? static_cast<int>(cw[P.bit_pos[i]]) : 0;
            err += est != tx;
        }
        err = wave_sum_fast(err);
        if (lane == 0 && err)
            atomicAdd(&misc[0], err);
        __syncthreads();
        if (tid == 0)
            a.bit_errors[frame] = static_cast<uint32_t>(misc[0]);
    }
}
} // namespace

// LDS-resident codes with check nodes up to degree 8, nc <= 8192 and 4 nnz >= 8 nc; a.ws_hb [n][nc] bytes must be set
bool fast_mode_supported(const DevPlan &p, int max_cn_degree)
{
    return max_cn_degree <= 8 && p.nc <= 8192 && 4ll * p.nnz >= 8ll * p.nc && p.vn_work_stride <= 8 && p.nnz < 65536 &&
           (static_cast<size_t>((p.nnz + 3) & ~3) + p.nc) * 4 + 64 <= 64 * 1024;
}

int launch_decode_fast(const DecodeArgs &a, int max_cn_degree, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    if (!fast_mode_supported(a.plan, max_cn_degree) || !a.ws_hb)
        return hipErrorInvalidValue;
    const uint32_t lds = (static_cast<uint32_t>((a.plan.nnz + 3) & ~3) + a.plan.nc) * 4u;
    void (*k)(const DecodeArgs) = a.llr_out ? decode_fast_kernel<true> : decode_fast_kernel<false>;
    hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(a.n_frames)), dim3(kFastThreads), lds, static_cast<hipStream_t>(stream), a);
    return hipGetLastError();
}

} // namespace ldpc_amd
