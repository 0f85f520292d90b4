// kernels_bec.hip — erasure decoding (BASELINE config 5, BEC) bit-sliced over 32 frames.
//
// The erasure decoder's alphabet is {0, 1, 'E'} and its two updates are closed forms over a node's other edges
// (src/decoding/decoder.cpp:96-186, decoder.h:145-155; kernels.hip states them): integer work, two bits per message.  One
// workgroup therefore decodes THIRTY-TWO consecutive frames at once: every message, symbol and output is a pair of 32-bit
// words in LDS — bit f of E = "erased in frame f", bit f of V = the value where it is not (V & E = 0 throughout) — and a
// lane that visits a node updates it for all 32 frames with a dozen bitwise instructions.  (32, not 64: the vector ALU is
// 32 bits wide, so a 64-bit word buys nothing per instruction; with 32 frames a group's state is 52 KB for h.txt and three
// groups share a CU — one's channel prologue, which waits for memory, runs under the others' passes — and the group's pass
// count, the maximum over its frames, is smaller.  Measured: 0.68 ms per 65 536 frames with 64-frame groups of 1024 threads.)  Per frame the results are
// exactly the byte-per-message kernel's (bec_kernel, kernels.hip — which remains for codes whose sliced state exceeds LDS):
//   check node, edge j     'E' if another input is erased, else the xor of the others.  Erased inputs are counted up to two
//                          bitwise (c0 = one or more, c1 = two or more): "no OTHER input erased" = ~c1 & (~c0 | E_j);
//   variable node          received symbol known: every output and the node's value are the transmitted bit x;
//     erased, degree >= 3  output x if another input equals x, else 'E' (inputs equal to x counted up to two the same way);
//     erased, degree 2     the other input unchanged; degree 1: the input is the node's value, the output 'E' — or 0 with
//                          deg1_compat (what the reference's out-of-bounds read yields);
//   early termination      per frame: a frame whose pass left no erasure stops counting and its outputs are frozen (the other
//                          frames of the group go on; the group ends when its last frame does or at the iteration limit);
//                          the iteration count of a frame = the passes after which it still held an erasure (decoder.cpp:169-186).
// Layout in LDS (4-byte words): ME[nnz] MV[nnz] (messages, check-node-major slots as in DevPlan), SE[nc] X[nc] (received
// symbol erased; transmitted bit), LE[nc] LV[nc] (the node's value, mLLROut), then the variable nodes' slot table (u16[nnz]).
// 512 threads per group.  The channel (channel.cpp:199-229) is part of the prologue: thread i turns the raw 64-bit draws of
// transmitted bit i of the 32 frames into one erasure word (loads coalesced over i).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "device_math.hpp"
#include "kernels.hpp"

namespace ldpc_amd
{

namespace
{

constexpr int kBecThreads = 512, kBecWaves = kBecThreads / 64;
constexpr int kBecFrames = 32; // frames per group = bits per word
#ifndef LDPC_AMD_BEC_WIDE
#define LDPC_AMD_BEC_WIDE 8
#endif
constexpr int kBecWideDegree = LDPC_AMD_BEC_WIDE; // variable nodes from this degree on are handled by four lanes each
using word_t = uint32_t;
constexpr uint8_t kErasureSym = 'E'; // functions.h:105

__device__ __forceinline__ word_t wave_or(word_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v |= static_cast<word_t>(__shfl_xor(static_cast<int>(v), o, 64));
    return v;
}

// bit f of the result = bit_of_frame(f) for the group's frames, 0 beyond them.  Written so that the 64 loads behind it are
// independent and unrolled (a frame index beyond the group reads the group's last frame and is masked out).
template <class F>
__device__ __forceinline__ word_t slice(int nf, word_t valid, F bit_of_frame)
{
    word_t w = 0;
#pragma unroll 8
    for (int f = 0; f < kBecFrames; ++f)
        w |= static_cast<word_t>(bit_of_frame(f < nf ? f : nf - 1) ? 1 : 0) << f;
    return w & valid;
}

// (pinned at six waves per SIMD: three groups of eight waves per CU is what the LDS allows as well)
__global__ __launch_bounds__(kBecThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void bec_sliced_kernel(const BecArgs a)
{
    extern __shared__ word_t ldsw[];
    __shared__ word_t still[2];
    __shared__ uint32_t errs[kBecFrames];
    const DevPlan &P = a.plan;
    const int nnz = P.nnz, nc = P.nc, nct = P.nct;
    word_t *ME = ldsw, *MV = ME + nnz, *SE = MV + nnz, *X = SE + nc, *LE = X + nc, *LV = LE + nc;
    uint16_t *slot = reinterpret_cast<uint16_t *>(LV + nc);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t f0 = static_cast<uint64_t>(blockIdx.x) * kBecFrames;
    const int nf = static_cast<int>(std::min<uint64_t>(kBecFrames, a.n_frames - f0)); // frames of this group
    const word_t valid = nf == kBecFrames ? ~word_t(0) : ((word_t(1) << nf) - 1);
    const uint8_t *cw = a.codeword ? a.codeword + f0 * nc : nullptr;

    if (tid < 2)
        still[tid] = 0;
    if (tid < kBecFrames)
        errs[tid] = 0;
    for (int e = tid; e < nnz; e += kBecThreads)
        slot[e] = static_cast<uint16_t>(P.vn_slot[e]);

    // ---- transmitted bits and channel: channel.cpp:199-229.  LV holds the received VALUES until the messages are initialised ----
    for (int r = tid; r < nc; r += kBecThreads)
    {
        word_t x = 0;
        if (cw)
        {
            const uint8_t *c = cw + P.rank_col[r];
            x = slice(nf, valid, [&](int f) { return c[static_cast<size_t>(f) * nc] != 0; });
        }
        X[r] = x;
        LE[r] = 0;
    }
    if (a.raw)
    {
        for (int r = tid; r < nc; r += kBecThreads)
        {
            const uint8_t k = P.rank_kind[r];
            if (k == 0)
                continue; // transmitted: below
            word_t se = 0, sv = 0;
            if (k == 1)
                se = ~word_t(0);
            else if (k == 2)
            {
                // channel.cpp:222 indexes the transmitted-symbol vector by the COLUMN index
                const uint32_t col = P.rank_col[r];
                if (col < static_cast<uint32_t>(nct) && cw)
                {
                    const uint8_t *c = cw + P.bit_pos[col];
                    sv = slice(nf, valid, [&](int f) { return c[static_cast<size_t>(f) * nc] != 0; });
                }
            } // k == 3: never written by the channel: the decoder's initial zero, a known 0 bit
            SE[r] = se, LV[r] = sv;
        }
        const uint64_t *raw = a.raw + f0 * static_cast<uint64_t>(nct);
        for (int i = tid; i < nct; i += kBecThreads)
        {
            word_t xb = 0;
            const double eps = a.eps;
            const word_t se = slice(nf, valid, [&](int f) { return canonical(raw[static_cast<size_t>(f) * nct + i]) < eps; });
            if (cw)
            {
                const uint8_t *c = cw + P.bit_pos[i];
                xb = slice(nf, valid, [&](int f) { return c[static_cast<size_t>(f) * nc] != 0; });
            }
            const uint32_t r = P.tx_rank[i];
            SE[r] = se, LV[r] = xb & ~se;
        }
    }
    else
    {
        const uint8_t *in = a.symbols + f0 * nc;
        for (int r = tid; r < nc; r += kBecThreads)
        {
            const uint8_t *s = in + P.rank_col[r];
            const word_t se = slice(nf, valid, [&](int f) { return s[static_cast<size_t>(f) * nc] == kErasureSym; });
            const word_t sv = slice(nf, valid, [&](int f) { return s[static_cast<size_t>(f) * nc] != 0; }) & ~se;
            SE[r] = se, LV[r] = sv;
        }
    }
    __syncthreads();
    if (a.llr_in_dump)
    {
        double *o = a.llr_in_dump + f0 * nc;
        for (int r = tid; r < nc; r += kBecThreads)
        {
            const word_t se = SE[r], sv = LV[r];
            double *d = o + P.rank_col[r];
            for (int f = 0; f < nf; ++f)
                d[static_cast<size_t>(f) * nc] = (se >> f) & 1 ? static_cast<double>(kErasureSym) : static_cast<double>((sv >> f) & 1);
        }
    }
    // ---- v2c init: decoder.cpp:96-99 ----
    for (int b = wave; b < P.n_vn_blocks; b += kBecWaves)
    {
        const auto t = uniform_table(reinterpret_cast<const uint32_t *>(P.vn_blocks + b));
        const uint32_t idx_off = t[0], first = t[1], cd = t[2];
        const int count = static_cast<int>(cd & 0xFFFFu), degree = static_cast<int>(cd >> 16);
        if (lane < count)
        {
            const word_t se = SE[first + lane], sv = LV[first + lane];
            for (int p = 0; p < degree; ++p)
            {
                const uint32_t s = slot[idx_off + lane + p * count];
                ME[s] = se, MV[s] = sv;
            }
        }
    }
    __syncthreads();
    for (int r = tid; r < nc; r += kBecThreads)
        LV[r] = 0; // mLLROut starts zeroed (the received values it held have gone into the messages)
    __syncthreads();

    // The wave's work, fixed for the whole decode, in the LANES of a few registers (lane k: its k-th item; read back in the
    // passes as scalars, v_readlane): a descriptor fetched by a scalar load from device memory in every pass is a round trip
    // the pass waits for, and at a few hundred instructions per wave and pass those round trips were most of the kernel
    // (0.63 ms per 65 536 frames with six of them per wave and pass, 0.77 with thirty-six).
    // Check-node blocks wave, wave + 8, ...; variable-node items dealt in order: a block of degree < kBecWideDegree is one
    // item, a wider one is a unit of 16 nodes per item, FOUR lanes per node (lane l of a quad takes the edges p = l, l + 4,
    // ...): the two degree-15 blocks of h.txt were one wave each at 15 edges per lane while six waves waited at the barrier.
    uint32_t c_off = 0, c_cd = 0, v_off = 0, v_first = 0, v_cd = 0;
    int v_unit = -1, n_cn = 0, n_vn = 0;
    for (int b = wave; b < P.n_cn_blocks; b += kBecWaves, ++n_cn)
    {
        const auto t = uniform_table(reinterpret_cast<const uint32_t *>(P.cn_blocks + b));
        if (lane == n_cn)
            c_off = t[0], c_cd = t[1];
    }
    {
        int item = 0;
        for (int b = 0; b < P.n_vn_blocks; ++b)
        {
            const auto t = uniform_table(reinterpret_cast<const uint32_t *>(P.vn_blocks + b));
            const uint32_t idx_off = t[0], first = t[1], cd = t[2];
            const int count = static_cast<int>(cd & 0xFFFFu), degree = static_cast<int>(cd >> 16);
            const int units = degree >= kBecWideDegree ? (count + 15) / 16 : 1;
            for (int u = 0; u < units; ++u, ++item)
                if (item % kBecWaves == wave)
                {
                    if (lane == n_vn)
                        v_off = idx_off, v_first = first, v_cd = cd, v_unit = degree >= kBecWideDegree ? u : -1;
                    ++n_vn;
                }
        }
    }
    word_t act = a.iterations > 0 ? valid : 0; // frames still decoding (uniform)
    uint32_t my_iters = 0;                      // thread f < 32: iteration count of frame f
    const word_t compat = a.deg1_compat ? ~word_t(0) : 0;
    for (uint32_t I = 0; I < a.iterations && act; ++I)
    {
        // ---- CN update: decoder.cpp:105-123 ----
        for (int k = 0; k < n_cn; ++k)
        {
            const uint32_t off = __builtin_amdgcn_readlane(c_off, k), cd = __builtin_amdgcn_readlane(c_cd, k);
            const int count = static_cast<int>(cd & 0xFFFFu), degree = static_cast<int>(cd >> 16);
            if (lane < count)
            {
                word_t *me = ME + off + lane, *mv = MV + off + lane;
                // up to eight inputs stay in registers between the two sweeps (wave-uniform degree: the guards are scalar)
                auto small = [&]<int D>(std::integral_constant<int, D>) {
                    word_t e[D], v[D], c0 = 0, c1 = 0, xa = 0;
#pragma unroll
                    for (int j = 0; j < D; ++j)
                        e[j] = me[j * count], v[j] = mv[j * count];
#pragma unroll
                    for (int j = 0; j < D; ++j)
                    {
                        c1 |= c0 & e[j];
                        c0 |= e[j];
                        xa ^= v[j];
                    }
#pragma unroll
                    for (int j = 0; j < D; ++j)
                    {
                        const word_t alone = ~c1 & (~c0 | e[j]); // no OTHER input erased
                        me[j * count] = ~alone;
                        mv[j * count] = (xa ^ v[j]) & alone;
                    }
                };
                switch (degree)
                {
                case 2: small(std::integral_constant<int, 2>{}); break;
                case 3: small(std::integral_constant<int, 3>{}); break;
                case 4: small(std::integral_constant<int, 4>{}); break;
                case 5: small(std::integral_constant<int, 5>{}); break;
                case 6: small(std::integral_constant<int, 6>{}); break;
                case 7: small(std::integral_constant<int, 7>{}); break;
                case 8: small(std::integral_constant<int, 8>{}); break;
                default:
                {
                    word_t c0 = 0, c1 = 0, xa = 0;
                    for (int j = 0; j < degree; ++j)
                    {
                        const word_t e = me[j * count];
                        c1 |= c0 & e;
                        c0 |= e;
                        xa ^= mv[j * count];
                    }
                    for (int j = 0; j < degree; ++j)
                    {
                        const word_t e = me[j * count], v = mv[j * count];
                        const word_t alone = ~c1 & (~c0 | e);
                        me[j * count] = ~alone;
                        mv[j * count] = (xa ^ v) & alone;
                    }
                }
                }
            }
        }
        __syncthreads();
        // ---- VN update: decoder.cpp:126-167 ----
        word_t any_e = 0;
        for (int k = 0; k < n_vn; ++k)
        {
            const uint32_t idx_off = __builtin_amdgcn_readlane(v_off, k), first = __builtin_amdgcn_readlane(v_first, k),
                           cd = __builtin_amdgcn_readlane(v_cd, k);
            const int u = __builtin_amdgcn_readlane(v_unit, k);
            const int count = static_cast<int>(cd & 0xFFFFu), degree = static_cast<int>(cd >> 16);
            if (u >= 0)
            {
                // a unit of a wide block: the counts up to two ("an input equals x": h0 one or more, h1 two or more) of a
                // quad's lanes combine by two exchanges within the quad
                const int n = u * 16 + (lane >> 2), sub = lane & 3;
                const bool on_node = n < count;
                const int r = static_cast<int>(first) + (on_node ? n : 0);
                const word_t se = SE[r], x = X[r];
                const uint16_t *idx = slot + idx_off + (on_node ? n : 0);
                word_t h0 = 0, h1 = 0;
                if (on_node)
                    for (int p = sub; p < degree; p += 4)
                    {
                        const uint32_t s = idx[p * count];
                        const word_t hit = ~ME[s] & ~(MV[s] ^ x);
                        h1 |= h0 & hit;
                        h0 |= hit;
                    }
                // combine the quad: first with the neighbour (lane ^ 1), then with the other pair (lane ^ 2)
#pragma unroll
                for (int o = 1; o <= 2; o <<= 1)
                {
                    const word_t g0 = static_cast<word_t>(__shfl_xor(static_cast<int>(h0), o, 64));
                    const word_t g1 = static_cast<word_t>(__shfl_xor(static_cast<int>(h1), o, 64));
                    h1 |= g1 | (h0 & g0);
                    h0 |= g0;
                }
                if (on_node)
                {
                    for (int p = sub; p < degree; p += 4)
                    {
                        const uint32_t s = idx[p * count];
                        const word_t hit = ~ME[s] & ~(MV[s] ^ x);
                        const word_t other = h1 | (h0 & ~hit); // another input equals x
                        const word_t e = se & ~other;
                        ME[s] = e, MV[s] = x & ~e;
                    }
                    if (sub == 0)
                    {
                        const word_t le = se & ~h0, lv = x & ~le;
                        LE[r] = (LE[r] & ~act) | (le & act);
                        LV[r] = (LV[r] & ~act) | (lv & act);
                        any_e |= le & act;
                    }
                }
                continue;
            }
            if (lane < count)
            {
                const int r = static_cast<int>(first) + lane;
                const word_t se = SE[r], x = X[r];
                const uint16_t *idx = slot + idx_off + lane;
                word_t le, lv;
                if (degree == 0)
                {
                    le = se, lv = x & ~se; // no edges: the erasure stays
                }
                else if (degree == 1)
                {
                    const uint32_t s0 = idx[0];
                    const word_t e0 = ME[s0], v0 = MV[s0];
                    ME[s0] = se & ~compat;
                    MV[s0] = x & ~se;
                    le = se & e0, lv = (x & ~se) | (v0 & se);
                }
                else if (degree == 2)
                {
                    const uint32_t s0 = idx[0], s1 = idx[count];
                    const word_t e0 = ME[s0], v0 = MV[s0], e1 = ME[s1], v1 = MV[s1];
                    ME[s0] = se & e1, MV[s0] = (x & ~se) | (v1 & se);
                    ME[s1] = se & e0, MV[s1] = (x & ~se) | (v0 & se);
                    const word_t hit = (~e0 & ~(v0 ^ x)) | (~e1 & ~(v1 ^ x));
                    le = se & ~hit, lv = x & ~le;
                }
                else
                {
                    word_t h0 = 0, h1 = 0;
                    for (int p = 0; p < degree; ++p)
                    {
                        const uint32_t s = idx[p * count];
                        const word_t hit = ~ME[s] & ~(MV[s] ^ x);
                        h1 |= h0 & hit;
                        h0 |= hit;
                    }
                    for (int p = 0; p < degree; ++p)
                    {
                        const uint32_t s = idx[p * count];
                        const word_t hit = ~ME[s] & ~(MV[s] ^ x);
                        const word_t other = h1 | (h0 & ~hit); // another input equals x
                        const word_t e = se & ~other;
                        ME[s] = e, MV[s] = x & ~e;
                    }
                    le = se & ~h0, lv = x & ~le;
                }
                LE[r] = (LE[r] & ~act) | (le & act); // frames that have finished keep their outputs
                LV[r] = (LV[r] & ~act) | (lv & act);
                any_e |= le & act;
            }
        }
        any_e = wave_or(any_e);
        if (lane == 0 && any_e)
            atomicOr(&still[I & 1], any_e);
        __syncthreads();
        // early termination when no erasure is left (decoder.cpp:169-186): the frames that go on, and count this pass
        const word_t on = a.early_term ? (act & still[I & 1]) : act;
        if (tid == 0)
            still[(I + 1) & 1] = 0; // (next written after the next pass's first barrier)
        if (tid < kBecFrames)
            my_iters += static_cast<uint32_t>((on >> tid) & 1);
        act = on;
    }
    __syncthreads();

    if (tid < nf && a.iters)
        a.iters[f0 + tid] = my_iters;
    const bool ran = a.iterations > 0;
    // mCO: decoder.cpp:137,165 — the true bit, or 1 when the VN is still erased (-gf2 is always 1)
    if (a.hard)
    {
        uint8_t *h = a.hard + f0 * nc;
        for (int r = tid; r < nc; r += kBecThreads)
        {
            const word_t hb = ran ? (LE[r] | X[r]) : 0;
            uint8_t *d = h + P.rank_col[r];
            for (int f = 0; f < nf; ++f)
                d[static_cast<size_t>(f) * nc] = static_cast<uint8_t>((hb >> f) & 1);
        }
    }
    if (a.llr_out)
    {
        double *o = a.llr_out + f0 * nc;
        for (int r = tid; r < nc; r += kBecThreads)
        {
            const word_t le = LE[r], lv = LV[r];
            double *d = o + P.rank_col[r];
            for (int f = 0; f < nf; ++f)
                d[static_cast<size_t>(f) * nc] = (le >> f) & 1 ? static_cast<double>(kErasureSym) : static_cast<double>((lv >> f) & 1);
        }
    }
    if (a.bit_errors)
    {
        // a transmitted bit is in error where the node is still erased and the bit is 0 (hard decision 1); the words of the
        // positions go to the (now free) message array, then the threads count them frame by frame
        word_t *ew = ME;
        for (int i = tid; i < P.n_bitpos; i += kBecThreads)
        {
            const uint32_t r = P.tx_rank[i];
            ew[i] = ran ? (LE[r] & ~X[r]) : X[r]; // (no iteration ran: every decision is 0, wrong where the bit is 1)
        }
        __syncthreads();
        // thread (part, f): frame f over every 16th position (two parts per wave)
        uint32_t n = 0;
        const int fr = tid & (kBecFrames - 1), part = tid / kBecFrames, parts = kBecThreads / kBecFrames;
        for (int i = part; i < P.n_bitpos; i += parts)
            n += static_cast<uint32_t>((ew[i] >> fr) & 1);
        if (n)
            atomicAdd(&errs[fr], n);
        __syncthreads();
        if (tid < nf)
            a.bit_errors[f0 + tid] = errs[tid];
    }
}

} // namespace

uint32_t bec_sliced_lds_bytes(const DevPlan &p)
{
    return static_cast<uint32_t>(8ull * p.nnz + 16ull * p.nc + 2ull * p.nnz + 16);
}

bool bec_sliced_fits(const DevPlan &p)
{
    // (n_bitpos words reuse the message array; block descriptors as kernels.hpp lays them out: 8 and 12 bytes)
    // (a wave's work list lives in the 64 lanes of a register: check-node blocks / 8, variable-node items / 8 <= 64)
    return bec_sliced_lds_bytes(p) <= 160u * 1024u - 1024u && p.n_bitpos <= p.nnz && p.nnz > 0 && p.nnz < 65536 &&
           p.n_cn_blocks <= 512 && 4 * p.n_vn_blocks <= 512;
}

int launch_bec_sliced(const BecArgs &a, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    static_assert(sizeof(CnBlock) == 8 && sizeof(VnBlock) == 12, "block descriptors are read as 2 / 3 scalar words");
    const uint32_t lds = bec_sliced_lds_bytes(a.plan);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bec_sliced_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(bec_sliced_kernel, dim3(static_cast<unsigned>((a.n_frames + kBecFrames - 1) / kBecFrames)), dim3(kBecThreads), lds,
                       static_cast<hipStream_t>(stream), a);
    return hipGetLastError();
}

} // namespace ldpc_amd
