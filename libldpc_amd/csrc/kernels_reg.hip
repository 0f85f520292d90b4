// kernels_reg.hip — register-resident BP / min-sum decoder for codes whose messages do not fit LDS four
// frames at a time (BASELINE config 4: (3,6)-regular n=8192, 196 KB of fp64 messages per frame).
//
// One workgroup of NT threads decodes one frame (NT = 512: two frames per CU, so that one frame's LDS
// exchange overlaps the other's arithmetic; NT = 1024: one).  The frame's messages live in the register file:
// thread (wave, lane) owns the check nodes of CN blocks k*(NT/64) + wave (k < KC) and holds their KC x MAXD messages
// in m[k][j] for the whole decode, so the check-node pass — 80 % of the arithmetic — touches no memory at all.
// The variable-node side is reached through an LDS mailbox laid out VN-block-major: CN threads scatter c2v,
// VN threads read their column contiguously ([position][lane]: conflict-free), form the APP in column file
// order, write v2c and the hard decision back in place, CN threads gather.  A code whose edges exceed the
// mailbox (160 KB = 18 176 entries of 8+1 bytes) is exchanged in rounds of VN blocks.  Input LLRs and per-VN
// hard decisions sit in device memory (read / written once per VN per iteration, coalesced).
//
// Same arithmetic, same order as the LDS-resident kernel (kernels.hip) and as the reference:
//   decode loop src/decoding/decoder.cpp:11-78, CN recursion :31-44 (device_cn.hpp), VN sum :50-56 in column
//   file order, syndrome src/decoding/decoder.h:47-64, channels src/sim/channel.cpp (device_channel.hpp).
#include <hip/hip_runtime.h>

#include <utility>

#include "device_channel.hpp"
#include "device_cn.hpp"
#include "device_math.hpp"
#include "kernels.hpp"

namespace ldpc_amd
{

namespace
{

// (the shared-reciprocal form of detmath.h belongs to the LDS-resident decoder: its range check rides on that kernel's
// check-node-first loop; here every output is divided separately)
// SH6: first launch of three — check nodes of degree 6 share reciprocals (detmath.h, dm_cn6_shared); their products' range
// check joins `escaped`, voted on after the variable-node pass that follows
template <bool MINSUM, bool RATIO, int MAXD, bool SH6>
__device__ __forceinline__ void cn_regs(double (&m)[MAXD], int degree, uint32_t *escaped)
{
    // wave-uniform degree: one fully unrolled recursion per width
#define LDPC_CASE(D)                         \
    case D:                                  \
    {                                        \
        double v[D];                         \
        _Pragma("unroll") for (int j = 0; j < D; ++j) v[j] = m[j]; \
        if constexpr (RATIO)                 \
            cn_ratio<D, false, SH6>(v, nullptr, escaped); \
        else                                 \
            cn_core<D, MINSUM>(v);           \
        _Pragma("unroll") for (int j = 0; j < D; ++j) m[j] = v[j]; \
        break;                               \
    }
    switch (degree)
    {
        LDPC_CASE(2)
        LDPC_CASE(3)
    default:
        if constexpr (MAXD >= 4)
            switch (degree)
            {
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
        break;
    }
#undef LDPC_CASE
}

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// RATIO: the likelihood-ratio form of the sum-product iteration, exactly as in kernels.hip (v2c = rho, c2v = lambda,
// input LLRs kept as lambda; frames that leave the representable box go to a.redo_list).
template <bool MINSUM, bool WANT_LLR, int NT, int KC, int MAXD, bool RATIO, bool SH6 = false>
__global__ __launch_bounds__(NT) void decode_reg_kernel(const DecodeArgs a, const DevRegPlan R)
{
    static_assert(!(RATIO && MINSUM), "the ratio form is a sum-product form");
    static_assert(RATIO || !SH6, "shared reciprocals belong to the ratio form");
    constexpr int kRegWaves = NT / 64;
    extern __shared__ double mb[]; // mailbox: mb_doubles doubles, then mb_doubles hard-bit bytes
    __shared__ int misc[4];
    uint8_t *hbm = reinterpret_cast<uint8_t *>(mb + R.mb_doubles);
    const DevPlan &P = a.plan;
    const int nc = P.nc;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_amdgcn_s_setprio(3); // ahead of the noise generator's waves in the SIMD's instruction arbitration (kernels.hip)
    uint64_t frame = blockIdx.x;
    if (a.redo_count_in) // second pass: only the frames the ratio form handed back
    {
        if (blockIdx.x >= *uniform_table(a.redo_count_in))
            return;
        frame = uniform_table(a.redo_list_in)[blockIdx.x];
    }
    double *llr = a.ws_llr + frame * nc;
    uint8_t *hard = a.ws_hb + frame * nc;
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;

    if (tid == 0)
        misc[0] = 0;
    channel_init<NT>(a, frame, llr, tid);
    __syncthreads();
    if (a.llr_in_dump)
    {
        double *o = a.llr_in_dump + frame * nc;
        for (int r = tid; r < nc; r += NT)
            o[P.rank_col[r]] = llr[r];
    }
    uint32_t escaped = 0; // RATIO: running maximum of dm_ratio_key over the frame's checked values (detmath.h)
    if constexpr (RATIO)
    {
        // input LLRs become lambda = e^-L in place (isolated variable nodes keep their LLR)
        for (int r = tid; r < nc; r += NT)
            if (P.rank_slot0[r] != kNoSlot)
            {
                const double L = llr[r];
                if (!(__builtin_fabs(L) <= DM_RATIO_LLR_LIMIT))
                    escaped = ~0u;
                llr[r] = dm_exp_clamped(0.0 - L);
            }
        __syncthreads();
    }

    double m[KC][MAXD];
    int deg[KC];
    bool have[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k)
    {
        deg[k] = R.cn_deg[k * kRegWaves + wave];
        have[k] = lane < static_cast<int>(R.cn_cnt[k * kRegWaves + wave]);
#pragma unroll
        for (int j = 0; j < MAXD; ++j)
            m[k][j] = 0.0;
    }
    const uint32_t *my_edge = R.cn_edge + tid;

    // ---- v2c initialisation (decoder.cpp:16-19): every edge starts with its VN's input LLR ----
    for (int r = 0; r < R.rounds; ++r)
    {
        for (uint32_t b = uniform_table(R.round_first)[r] + wave; b < uniform_table(R.round_first)[r + 1]; b += kRegWaves)
        {
            const RegVnBlock vb = load_block3(R.vn_blocks, b);
            if (lane < vb.count)
            {
                const double L = llr[vb.first + lane];
                const double v0 = RATIO ? dm_ratio_div(1.0, L) : L; // RATIO: L is lambda(L_ch), the first v2c is rho(L_ch)
                for (int p = 0; p < vb.degree; ++p)
                    mb[vb.mb_off + p * vb.count + lane] = v0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int j = 0; j < MAXD; ++j)
            {
                const uint32_t e = my_edge[(k * MAXD + j) * NT];
                if (e != kRegNoEdge && (e >> 28) == static_cast<uint32_t>(r))
                    m[k][j] = mb[e & 0x0FFFFFFFu];
            }
        __syncthreads();
    }

    double *out_llr = WANT_LLR ? a.llr_out + frame * nc : nullptr;
    uint32_t I = 0;
    while (I < a.iterations)
    {
        // ---- CN pass (decoder.cpp:25-45), entirely in registers ----
        // (a fold expression, not a loop: every m[k] must be a compile-time register row)
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
            ((have[Ks] ? cn_regs<MINSUM, RATIO, MAXD, SH6>(m[Ks], deg[Ks], &escaped) : void()), ...);
        }(std::make_integer_sequence<int, KC>{});

        int par[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k)
            par[k] = 0;
        for (int r = 0; r < R.rounds; ++r)
        {
            // c2v -> mailbox
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int j = 0; j < MAXD; ++j)
                {
                    const uint32_t e = my_edge[(k * MAXD + j) * NT];
                    if (e != kRegNoEdge && (e >> 28) == static_cast<uint32_t>(r))
                        mb[e & 0x0FFFFFFFu] = m[k][j];
                }
            __syncthreads();
            // ---- VN pass, APP and hard decision (decoder.cpp:48-64) on this round's VN blocks ----
            for (uint32_t b = uniform_table(R.round_first)[r] + wave; b < uniform_table(R.round_first)[r + 1]; b += kRegWaves)
            {
                const RegVnBlock vb = load_block3(R.vn_blocks, b);
                if (lane < vb.count)
                {
                    const int rank = vb.first + lane;
                    double *col = mb + vb.mb_off + lane;
                    uint8_t *hcol = hbm + vb.mb_off + lane;
                    if constexpr (RATIO)
                        if (vb.degree > 0)
                        {
                            if (vb.degree == 1)
                            {
                                // a leaf: its v2c is the channel ratio itself, the decision lambda(c2v) >= rho_ch
                                // (kernels.hip, vn_leaf_ratio)
                                const double c = col[0], rho = dm_ratio_div(1.0, llr[rank]);
                                const uint8_t lbit = c >= rho;
                                col[0] = rho;
                                hcol[0] = lbit;
                                hard[rank] = lbit;
                                if constexpr (WANT_LLR)
                                    out_llr[P.rank_col[rank]] = 0.0 - dm_log(dm_ratio_div(c, rho));
                                continue;
                            }
                            // lambda(total) = lambda(L_ch) * prod lambda(c2v_p), in column file order
                            double prod = llr[rank];
                            if (vb.degree <= 3)
                                for (int p = 0; p < vb.degree; ++p)
                                    prod *= col[p * vb.count];
                            else
                                for (int p = 0; p < vb.degree; ++p)
                                {
                                    prod *= col[p * vb.count];
                                    if (p % 3 == 2)
                                        DM_RATIO_TRACK(escaped, prod);
                                }
                            const uint8_t bit = prod >= 1.0; // total LLR <= 0
                            const double tot = dm_ratio_div(1.0, prod);   // rho(total)
                            for (int p = 0; p < vb.degree; ++p)
                            {
                                const double o = tot * col[p * vb.count]; // rho(total - c2v_p)
                                DM_RATIO_TRACK(escaped, o);
                                col[p * vb.count] = o;
                                hcol[p * vb.count] = bit;
                            }
                            hard[rank] = bit;
                            if constexpr (WANT_LLR)
                                out_llr[P.rank_col[rank]] = 0.0 - dm_log(prod);
                            continue;
                        }
                    double out = llr[rank];
                    for (int p = 0; p < vb.degree; ++p) // sequential sum in column file order
                        out += col[p * vb.count];
                    const uint8_t bit = out <= 0;
                    for (int p = 0; p < vb.degree; ++p)
                    {
                        col[p * vb.count] = out - col[p * vb.count];
                        hcol[p * vb.count] = bit;
                    }
                    hard[rank] = bit;
                    if constexpr (WANT_LLR)
                        out_llr[P.rank_col[rank]] = out;
                }
            }
            __syncthreads();
            // v2c (and the hard decision of the edge's VN) <- mailbox
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int j = 0; j < MAXD; ++j)
                {
                    const uint32_t e = my_edge[(k * MAXD + j) * NT];
                    if (e != kRegNoEdge && (e >> 28) == static_cast<uint32_t>(r))
                    {
                        m[k][j] = mb[e & 0x0FFFFFFFu];
                        par[k] ^= hbm[e & 0x0FFFFFFFu];
                    }
                }
            __syncthreads();
        }
        // ---- syndrome early termination (decoder.cpp:66-72, decoder.h:47-64) ----
        if constexpr (RATIO)
            if (__syncthreads_or(DM_RATIO_ESCAPED(escaped))) // checked before the syndrome: an escaped frame's hard decisions mean nothing
            {
                if (tid == 0)
                    a.redo_list[atomicAdd(a.redo_count, 1u)] = static_cast<uint32_t>(frame);
                return;
            }
        if (a.early_term)
        {
            int bad = 0;
#pragma unroll
            for (int k = 0; k < KC; ++k)
                bad |= have[k] ? par[k] : 0;
            if (!__syncthreads_or(bad))
                break;
        }
        ++I;
    }
    __syncthreads();

    // ---- outputs ----
    if (tid == 0 && a.iters)
        a.iters[frame] = I;
    const bool ran = a.iterations > 0;
    if (a.hard)
    {
        uint8_t *h = a.hard + frame * nc;
        for (int r = tid; r < nc; r += NT)
            h[P.rank_col[r]] = ran ? hard[r] : 0;
    }
    if constexpr (WANT_LLR)
    {
        if (!ran)
            for (int r = tid; r < nc; r += NT)
                out_llr[P.rank_col[r]] = 0.0;
    }
    if (a.bit_errors)
    {
        int err = 0;
        for (int i = tid; i < P.n_bitpos; i += NT)
        {
            int est = ran ? hard[P.tx_rank[i]] : 0;
            int tx = cw ? static_cast<int>(cw[P.bit_pos[i]]) : 0;
            err += est != tx;
        }
        err = wave_sum_i(err);
        if (lane == 0 && err)
            atomicAdd(&misc[0], err);
        __syncthreads();
        if (tid == 0)
            a.bit_errors[frame] = static_cast<uint32_t>(misc[0]);
    }
}

template <int NT, int KC, int MAXD>
int launch_reg(const DecodeArgs &a, const DevRegPlan &r, bool min_sum, void *stream)
{
    const bool want_llr = a.llr_out != nullptr;
    const bool ratio = a.redo_list != nullptr; // (with a list coming in as well: the second launch, outputs divided separately)
    if (ratio && (min_sum || !a.early_term || a.iterations == 0 || !a.redo_count))
        return hipErrorInvalidValue;
    void (*k)(const DecodeArgs, const DevRegPlan) = nullptr;
    if (min_sum)
        k = want_llr ? decode_reg_kernel<true, true, NT, KC, MAXD, false> : decode_reg_kernel<true, false, NT, KC, MAXD, false>;
    else if (ratio && !a.redo_count_in && MAXD >= 6)
        k = want_llr ? decode_reg_kernel<false, true, NT, KC, MAXD, true, MAXD >= 6> : decode_reg_kernel<false, false, NT, KC, MAXD, true, MAXD >= 6>;
    else if (ratio)
        k = want_llr ? decode_reg_kernel<false, true, NT, KC, MAXD, true> : decode_reg_kernel<false, false, NT, KC, MAXD, true>;
    else
        k = want_llr ? decode_reg_kernel<false, true, NT, KC, MAXD, false> : decode_reg_kernel<false, false, NT, KC, MAXD, false>;
    const uint32_t lds = r.mb_doubles * 9u;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(a.n_frames)), dim3(NT), lds, static_cast<hipStream_t>(stream), a, r);
    return hipGetLastError();
}

} // namespace

int launch_decode_reg(const DecodeArgs &a, const DevRegPlan &r, bool min_sum, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    if (!a.ws_llr || !a.ws_hb)
        return hipErrorInvalidValue;
#define LDPC_TILE(N, K, D)                      \
    if (r.nt == N && r.kc == K && r.maxd == D)  \
        return launch_reg<N, K, D>(a, r, min_sum, stream);
    LDPC_TILE(512, 8, 6)
    LDPC_TILE(512, 16, 4)
    LDPC_TILE(512, 4, 8)
    LDPC_TILE(1024, 4, 6)
    LDPC_TILE(1024, 8, 4)
    LDPC_TILE(1024, 2, 8)
#undef LDPC_TILE
    return hipErrorInvalidValue;
}

} // namespace ldpc_amd
