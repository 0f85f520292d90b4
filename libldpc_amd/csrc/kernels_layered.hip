// kernels_layered.hip — opt-in NON-PARITY modes 2 and 3: a LAYERED (row-serial) sum-product
// schedule, with binary32 or binary16 check-to-variable messages.  Results are NOT those of the reference, whose
// schedule is flooding (decoder.cpp:22-76): the modes exist for throughput studies, are never chosen by the library
// itself (ldpc_hip_set_fast_mode 2 / 3), and every parity test runs the binary64 flooding kernels.  What they cost or
// gain in error rate and iterations is measured, not asserted (tools/fast_mode_report.py, profiles/).
//
// Layered belief propagation keeps one total per variable node (its a-posteriori LLR) and visits the check nodes in
// sequence; a check node takes total - (its own previous message) from each neighbour, forms its new messages, and puts
// total - old + new back at once, so later check nodes of the same sweep already see it: about half the sweeps of the
// flooding schedule for the same error rate.  (The idea of processing the rows in layers is the one the reference's legacy
// simulator has in gpu/ldpc/ldpc.cpp:111-138; nothing of its code or layer construction is used.)
//
// Mapping, chosen for the schedule rather than inherited from the flooding kernels: ONE WAVE = ONE FRAME.  A sweep is a
// sequence of STEPS; a step is up to 64 check nodes of equal degree that share no variable node (plan.cpp,
// build_layer_plan: greedy packing), one per lane, so the lanes of a step never touch the same total and the steps of a
// sweep need no barrier at all — a wave's LDS operations execute in order.  Per frame LDS holds the totals (binary32) and
// the check-to-variable messages (binary32 or binary16, [step][edge][lane]: conflict-free): 20 KB or 12 KB for the n=1024
// code, seven or twelve frames per CU.  The check node itself is the ratio form of kernels_fast.hip ((a + b) / (1 + a b)
// with v_rcp_f32, forward/backward order), messages in log2 units clipped to +-kClip.
//
// Iteration count returned: sweeps completed before the sweep whose syndrome check passed (the reference's convention,
// decoder.cpp:21-22,74-77); the syndrome of the current decisions is taken after every sweep.
#include <hip/hip_runtime.h>

#include "device_channel.hpp"
#include "device_math.hpp"
#include "kernels.hpp"

namespace ldpc_amd
{

namespace
{
constexpr int kLayWaves = 1; // frames per workgroup: one wave, so that a CU takes as many frames as its LDS holds
constexpr int kLayThreads = 64 * kLayWaves;
constexpr float kClip = 40.0f; // |L2| <= 40: |LLR| <= 27.7
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ float l_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float l_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float l_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float l_lambda(float a, float b) { return (a + b) * l_rcp(__builtin_fmaf(a, b, 1.0f)); }
__device__ __forceinline__ float l_rho(float a, float b) { return __builtin_fmaf(a, b, 1.0f) * l_rcp(a + b); }
__device__ __forceinline__ float clip(float x) { return __builtin_fminf(__builtin_fmaxf(x, -kClip), kClip); }

template <typename M>
__device__ __forceinline__ float msg_load(const M *p)
{
    return static_cast<float>(*p);
}
template <typename M>
__device__ __forceinline__ void msg_store(M *p, float v)
{
    *p = static_cast<M>(v);
}

// one check node of degree D on this lane: tot[] = the frame's totals, pk = its neighbours' VN ranks (two per word),
// c2v[j * 64] = its messages
template <int D, typename M>
__device__ __forceinline__ void cn_layered(float *tot, M *c2v, const uint32_t (&pk)[4])
{
    uint32_t n[D];
    float t[D], v[D];
#pragma unroll
    for (int j = 0; j < D; ++j)
        n[j] = (j & 1) ? pk[j >> 1] >> 16 : pk[j >> 1] & 0xFFFFu;
#pragma unroll
    for (int j = 0; j < D; ++j)
    {
        t[j] = tot[n[j]] - msg_load(c2v + j * 64); // what the neighbour says without this node's last message
        v[j] = l_exp2(clip(t[j]));                 // as a likelihood ratio rho = 2^L2; only what ENTERS the check node is
    }                                              // clipped: the total keeps everything the other check nodes have said
    float o[D];
    if constexpr (D == 2)
    {
        o[0] = l_rcp(v[1]), o[1] = l_rcp(v[0]);
    }
    else
    {
        float F[D], B[D]; // partial results as rho, forward / backward (decoder.cpp:31-44)
        F[0] = v[0], B[D - 1] = v[D - 1];
#pragma unroll
        for (int j = 1; j <= D - 3; ++j)
            F[j] = l_rho(F[j - 1], v[j]);
#pragma unroll
        for (int j = D - 2; j >= 2; --j)
            B[j] = l_rho(B[j + 1], v[j]);
        o[0] = l_lambda(D > 3 ? B[2] : v[2], v[1]);
        o[D - 1] = l_lambda(D > 3 ? F[D - 3] : v[0], v[D - 2]);
#pragma unroll
        for (int j = 1; j < D - 1; ++j)
            o[j] = l_lambda(F[j - 1], B[j + 1]);
    }
#pragma unroll
    for (int j = 0; j < D; ++j)
    {
        const float m = clip(0.0f - l_log2(o[j])); // lambda = 2^-L2 -> the new message in log2 units
        M mm;
        msg_store(&mm, m);
        c2v[j * 64] = mm;
        tot[n[j]] = t[j] + msg_load(&mm); // the total carries exactly the message that was stored (binary16: its rounding)
    }
}

template <int D>
__device__ __forceinline__ uint32_t cn_parity(const float *tot, const uint32_t (&pk)[4])
{
    uint32_t p = 0;
#pragma unroll
    for (int j = 0; j < D; ++j)
    {
        const float x = tot[(j & 1) ? pk[j >> 1] >> 16 : pk[j >> 1] & 0xFFFFu];
        p ^= static_cast<uint32_t>(x <= 0.0f); // decoder.cpp:58: out <= 0 decides 1
    }
    return p;
}

template <typename M, bool WANT_LLR>
__global__ __launch_bounds__(kLayThreads) void decode_layered_kernel(const DecodeArgs a, const DevLayerPlan L)
{
    extern __shared__ double lds_d[];
    const DevPlan &P = a.plan;
    const int nc = P.nc;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t frame = static_cast<uint64_t>(blockIdx.x) * kLayWaves + wave;
    if (frame >= a.n_frames) // (no workgroup barrier anywhere below: a wave may leave)
        return;
    // this wave's LDS: the channel's binary64 LLRs first, then float tot[nc] followed by the messages in the same bytes
    double *region = lds_d + static_cast<size_t>(wave) * ((sizeof(M) == 2 ? L.region_bytes_half : L.region_bytes) / 8);
    double *llr = region;
    float *tot = reinterpret_cast<float *>(region);
    M *c2v = reinterpret_cast<M *>(tot + ((nc + 3) & ~3));
    const uint8_t *cw = a.codeword ? a.codeword + frame * nc : nullptr;

    channel_init<64>(a, frame, llr, lane); // binary64 channel + LLR initialisation, as everywhere (one wave's worth)
    __builtin_amdgcn_wave_barrier();
    if (a.llr_in_dump)
    {
        double *o = a.llr_in_dump + frame * nc;
        for (int r = lane; r < nc; r += 64)
            o[P.rank_col[r]] = llr[r];
    }
    // binary64 LLRs -> totals in log2 units, in place: element r is read as a double (bytes 8r..) and written as a float
    // (bytes 4r..), 64 elements per pass in ascending order — a pass only overwrites bytes that it or an earlier pass has read
    for (int r0 = 0; r0 < nc; r0 += 64)
    {
        const int r = r0 + lane;
        const double x = r < nc ? llr[r] : 0.0;
        __builtin_amdgcn_wave_barrier();
        if (r < nc)
            tot[r] = clip(static_cast<float>(x) * kLog2e);
        __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t s = lane; s < L.slots; s += 64)
        msg_store(c2v + s, 0.0f); // no message yet
    __builtin_amdgcn_wave_barrier();

    // A step's neighbour table (four words per lane: eight VN ranks, [step][word][lane]) is fetched one step AHEAD: the
    // steps of a sweep depend on each other through the totals, so a wave cannot overlap them, but the table does not
    // depend on anything — its global-memory latency hides under the step before.
    const auto steps = uniform_table(reinterpret_cast<const uint32_t *>(L.steps));
    const uint32_t *table = L.vn4 + lane;
    auto fetch = [&](uint32_t s, uint32_t (&pk)[4]) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
            pk[w] = table[(s * 4 + w) * 64];
    };
    uint32_t I = 0;
    bool converged = false;
    uint32_t nxt[4];
    fetch(0, nxt);
    while (I < a.iterations)
    {
        for (uint32_t s = 0; s < L.n_steps; ++s)
        {
            uint32_t cur[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
            fetch(s + 1 < L.n_steps ? s + 1 : 0, nxt); // (the sweep's last step fetches the first one's: the syndrome pass and the next sweep start there)
            const uint32_t off = steps[2 * s], cd = steps[2 * s + 1];
            const int count = cd & 0xFFFF, degree = cd >> 16;
            if (lane < count)
            {
                M *m = c2v + off + lane;
                switch (degree) // wave-uniform
                {
                case 2: cn_layered<2, M>(tot, m, cur); break;
                case 3: cn_layered<3, M>(tot, m, cur); break;
                case 4: cn_layered<4, M>(tot, m, cur); break;
                case 5: cn_layered<5, M>(tot, m, cur); break;
                case 6: cn_layered<6, M>(tot, m, cur); break;
                case 7: cn_layered<7, M>(tot, m, cur); break;
                case 8: cn_layered<8, M>(tot, m, cur); break;
                default: break;
                }
            }
            __builtin_amdgcn_wave_barrier(); // (LDS operations of a wave execute in order; this only stops the compiler)
        }
        // syndrome of the decisions after this sweep (decoder.cpp:66-72)
        if (a.early_term)
        {
            uint32_t bad = 0;
            for (uint32_t s = 0; s < L.n_steps; ++s)
            {
                uint32_t cur[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
                fetch(s + 1 < L.n_steps ? s + 1 : 0, nxt);
                const uint32_t cd = steps[2 * s + 1];
                const int count = cd & 0xFFFF, degree = cd >> 16;
                if (lane < count)
                {
                    switch (degree)
                    {
                    case 2: bad |= cn_parity<2>(tot, cur); break;
                    case 3: bad |= cn_parity<3>(tot, cur); break;
                    case 4: bad |= cn_parity<4>(tot, cur); break;
                    case 5: bad |= cn_parity<5>(tot, cur); break;
                    case 6: bad |= cn_parity<6>(tot, cur); break;
                    case 7: bad |= cn_parity<7>(tot, cur); break;
                    case 8: bad |= cn_parity<8>(tot, cur); break;
                    default: break;
                    }
                }
            }
            if (__ballot(bad != 0) == 0)
            {
                converged = true;
                break;
            }
        }
        ++I;
    }
    (void)converged;
    if (lane == 0 && a.iters)
        a.iters[frame] = I;
    const bool ran = a.iterations > 0;
    int err = 0;
    uint8_t *h = a.hard ? a.hard + frame * nc : nullptr;
    for (int r = lane; r < nc; r += 64)
    {
        const float x = tot[r];
        const uint8_t bit = ran ? static_cast<uint8_t>(x <= 0.0f) : 0; // mCO is still zero-initialised when no iteration ran
        if (h)
            h[P.rank_col[r]] = bit;
        if constexpr (WANT_LLR)
            a.llr_out[frame * nc + P.rank_col[r]] = ran ? static_cast<double>(x) * 0.6931471805599453 : 0.0;
    }
    if (a.bit_errors)
    {
        for (int i = lane; i < P.n_bitpos; i += 64)
        {
            const int est = ran ? static_cast<int>(tot[P.tx_rank[i]] <= 0.0f) : 0;
            const int tx = cw ? static_cast<int>(cw[P.bit_pos[i]]) : 0;
            err += est != tx;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            err += __shfl_xor(err, o, 64);
        if (lane == 0)
            a.bit_errors[frame] = static_cast<uint32_t>(err);
    }
}
} // namespace

int launch_decode_layered(const DecodeArgs &a, const DevLayerPlan &L, bool half_messages, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    if (!L.steps || L.region_bytes == 0)
        return hipErrorInvalidValue;
    const uint32_t lds = (half_messages ? L.region_bytes_half : L.region_bytes) * kLayWaves;
    void (*k)(const DecodeArgs, const DevLayerPlan) =
        half_messages ? (a.llr_out ? decode_layered_kernel<_Float16, true> : decode_layered_kernel<_Float16, false>)
                      : (a.llr_out ? decode_layered_kernel<float, true> : decode_layered_kernel<float, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess)
        return e;
    const unsigned blocks = static_cast<unsigned>((a.n_frames + kLayWaves - 1) / kLayWaves);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(kLayThreads), lds, static_cast<hipStream_t>(stream), a, L);
    return hipGetLastError();
}

} // namespace ldpc_amd
