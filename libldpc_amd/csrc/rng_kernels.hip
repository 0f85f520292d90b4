// rng_kernels.hip — the reference's noise source rebuilt for the GPU.
//
// The reference draws all channel noise from ONE sequential std::mt19937_64(seed) per thread
// (src/sim/channel.cpp:5-15,37-42), through libstdc++'s normal_distribution (Marsaglia polar
// method with rejection: a data-dependent number of draws per sample).  To keep frame f the f-th
// frame of that very stream while decoding tens of thousands of frames per launch, the stream is
// produced in three data-parallel steps:
//
//   mt_jump_kernel      start state of every chunk of the stream, by GF(2) jump-ahead
//                       (state_{n+J} = g_J(T) state_n; evaluated as a sliding XOR of sequence
//                       words selected by the coefficients of g_J = t^J mod charpoly),
//   mt_generate_kernel  one wave per chunk regenerates its 312-word state in LDS and writes the
//                       tempered 64-bit outputs, 512 B per store instruction,
//   polar_*_kernel      evaluates the polar acceptance test of every trial in parallel,
//                       prefix-sums the accept flags and compacts the accepted (u1,u2) pairs in
//                       stream order, so that normal number g is element g&1 of pair g>>1.
//
// The arithmetic of the acceptance test is device_math.hpp::polar_trial — the same code the
// decoder prologue uses to turn a pair into two normals.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "kernels.hpp"

namespace ldpc_amd
{

namespace
{

// one in-place regeneration of the 312-word window by one wave (see the hazard analysis in DESIGN.md):
// rounds of 64 consecutive k; every lane reads x[k], x[k+1], x[k+156] (mod 312) before any lane of the
// round writes, and a round only overwrites its own k range.
__device__ __forceinline__ void mt_regenerate(uint64_t *x, int lane)
{
#pragma unroll
    for (int r = 0; r < 5; ++r)
    {
        int k = r * 64 + lane;
        uint64_t v = 0;
        if (k < kMtN)
        {
            int k1 = k + 1 == kMtN ? 0 : k + 1;
            int km = k + 156 >= kMtN ? k + 156 - kMtN : k + 156;
            v = mt_twist(x[k], x[k1], x[km]);
        }
        __builtin_amdgcn_wave_barrier();
        if (k < kMtN)
            x[k] = v;
        __builtin_amdgcn_wave_barrier();
    }
}

// Three waves per chunk, kGenChunks chunks per workgroup.  The twist has 156-way parallelism: words 0..155 of the next
// window depend only on the current window, words 156..311 on the current window and the new words 0..155.
// Thread t < 156 of a chunk's three waves produces words t and t+156; reads and writes of a half are separated by
// workgroup barriers.  Several chunks share a workgroup so that a batch's 82 chunks occupy 21 compute units for the
// length of the serial chain and not 82: the register-resident decode kernel owns a whole CU per frame, and every CU
// that hosts a generator wave is lost to it for that long.
// (kGenChunks = 1 when the decode kernel shares its CUs anyway: packed chunks lengthen the serial chain, which then
// outlasts the headline decode kernel it runs under)
constexpr int kGenThreads = 192;
template <int kGenChunks>
__global__ __launch_bounds__(kGenThreads *kGenChunks) void mt_generate_kernel(const uint64_t *states, uint64_t *next_last,
                                                                              uint64_t *out, uint32_t chunk_words, uint32_t n_chunks)
{
    __shared__ uint64_t xs[kGenChunks][kMtN];
    const int t = threadIdx.x % kGenThreads, sub = threadIdx.x / kGenThreads;
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kGenChunks + sub;
    const bool live = c < n_chunks; // (the last workgroup may hold fewer chunks; its idle waves still meet the barriers)
    const bool act = live && t < 156;
    uint64_t *x = xs[sub];
    const uint64_t *s = states + c * kMtN;
    if (live)
        for (int k = t; k < kMtN; k += kGenThreads)
            x[k] = s[k];
    __syncthreads();
    uint64_t *o = out + c * chunk_words;
    const uint32_t blocks = chunk_words / kMtN;
    for (uint32_t b = 0; b < blocks; ++b)
    {
        uint64_t lo = 0, hi = 0, lo1 = 0, hi1 = 0;
        if (act)
        {
            lo = x[t], hi = x[t + 156];
            lo1 = x[t + 1];                           // t+1 <= 156
            hi1 = t + 157 < kMtN ? x[t + 157] : 0;    // word 312 wraps to the NEW word 0, taken below
            o[b * kMtN + t] = mt_temper(lo);
            o[b * kMtN + 156 + t] = mt_temper(hi);
        }
        __syncthreads();
        uint64_t nlo = 0;
        if (act)
        {
            nlo = mt_twist(lo, lo1, hi); // x[k+156] of the current window
            x[t] = nlo;
        }
        __syncthreads();
        if (act)
        {
            if (t == 155)
                hi1 = x[0]; // new word 0
            x[t + 156] = mt_twist(hi, hi1, nlo); // (k+156) mod 312 = k-156: the new low half
        }
        __syncthreads();
    }
    if (next_last && c + 1 == n_chunks) // the window after the last chunk = start state of the next chunk
        for (int k = t; k < kMtN; k += kGenThreads)
            next_last[k] = x[k];
}

// dst = window at position n+J given the window at n: word i of the new window is the XOR of the
// sequence words w[k+i] over all k with coefficient g_k = 1 (k < 19937).
//
// The sequence is never written to memory: the workgroup keeps two consecutive 312-word blocks of it
// in LDS (ring[0..311] = block b, ring[312..623] = block b+1).  In stage b thread i owns output word i
// and consumes the taps k in [312 b, 312 b + 312): w[k+i] = ring[k - 312 b + i], consecutive lanes read
// consecutive words.  Wave 0 then regenerates the next block and the ring advances.  The taps of a stage
// are taken eight at a time: eight independent LDS reads in flight, then eight XORs predicated on
// wave-uniform coefficient bits.
constexpr int kJumpThreads = 320;
constexpr int kJumpStages = 64;      // 64 * 312 = 19968 >= 19937 taps
constexpr int kPolyWords = 320;      // 312 coefficient words + zero padding read by the last stage

__global__ __launch_bounds__(kJumpThreads) void mt_jump_kernel(const uint64_t *src, uint64_t *dst,
                                                               const uint64_t *poly)
{
    __shared__ uint64_t x[kMtN];        // generator state = newest block
    __shared__ uint64_t ring[2 * kMtN];
    __shared__ uint64_t g[kPolyWords];
    const int tid = threadIdx.x;
    const uint64_t t = blockIdx.x;
    for (int k = tid; k < kPolyWords; k += kJumpThreads)
        g[k] = poly[k];
    if (tid < kMtN)
    {
        uint64_t v = src[t * kMtN + tid];
        x[tid] = v;
        ring[tid] = v;
    }
    __syncthreads();
    if (tid < 64)
        mt_regenerate(x, tid);
    __syncthreads();
    if (tid < kMtN)
        ring[kMtN + tid] = x[tid];
    __syncthreads();

    uint64_t acc = 0;
    for (int b = 0; b < kJumpStages; ++b)
    {
        if (tid < kMtN)
        {
            const int k0 = b * kMtN;
            for (int kk = 0; kk < kMtN; kk += 8)
            {
                // eight coefficient bits starting at k0 + kk (wave-uniform)
                const int k = k0 + kk;
                const int w = k >> 6, sh = k & 63;
                uint64_t lo = g[w] >> sh;
                uint64_t hi = sh ? (g[w + 1] << (64 - sh)) : 0;
                uint32_t bits = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(lo | hi))) & 0xFFu;
                if (bits == 0)
                    continue;
                uint64_t v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = ring[kk + j + tid];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (bits >> j & 1)
                        acc ^= v[j];
            }
        }
        __syncthreads();
        // advance: block b+1 becomes the low half, wave 0 produces block b+2
        if (tid < 64)
            mt_regenerate(x, tid);
        uint64_t up = tid < kMtN ? ring[kMtN + tid] : 0;
        __syncthreads();
        if (tid < kMtN)
        {
            ring[tid] = up;
            ring[kMtN + tid] = x[tid];
        }
        __syncthreads();
    }
    if (tid < kMtN)
        dst[t * kMtN + tid] = acc;
}

// ---- polar acceptance scan -------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanIters = kScanBlock / kScanThreads; // trials per thread

__device__ __forceinline__ bool trial_accepted(const uint64_t *raw, uint64_t t, uint64_t n_trials)
{
    if (t >= n_trials)
        return false;
    const ulonglong2 u = *reinterpret_cast<const ulonglong2 *>(raw + 2 * t);
    return polar_trial(u.x, u.y).accepted();
}

__global__ __launch_bounds__(kScanThreads) void polar_count_kernel(const uint64_t *raw, uint64_t n_trials,
                                                                   uint32_t *block_counts)
{
    __shared__ int total;
    if (threadIdx.x == 0)
        total = 0;
    __syncthreads();
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanBlock;
    int cnt = 0;
#pragma unroll
    for (int it = 0; it < kScanIters; ++it)
        cnt += trial_accepted(raw, base + it * kScanThreads + threadIdx.x, n_trials);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(&total, cnt);
    __syncthreads();
    if (threadIdx.x == 0)
        block_counts[blockIdx.x] = static_cast<uint32_t>(total);
}

// single-workgroup exclusive scan of the block counts
__global__ __launch_bounds__(1024) void polar_offsets_kernel(const uint32_t *block_counts, uint32_t n_blocks,
                                                             uint64_t *block_offsets, uint64_t want,
                                                             ScanResult *result)
{
    __shared__ uint64_t part[1024];
    const int tid = threadIdx.x;
    const uint32_t per = (n_blocks + 1023) / 1024;
    const uint32_t lo = tid * per, hi = min(lo + per, n_blocks);
    uint64_t s = 0;
    for (uint32_t b = lo; b < hi; ++b)
        s += block_counts[b];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) // Hillis-Steele inclusive scan
    {
        uint64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint64_t run = tid ? part[tid - 1] : 0;
    for (uint32_t b = lo; b < hi; ++b)
    {
        block_offsets[b] = run;
        run += block_counts[b];
    }
    if (tid == 1023)
    {
        result->accepted = part[1023];
        result->enough = part[1023] >= want;
        if (want == 0)
            result->trials_used = 0;
    }
}

// Registers: the kernel runs under the decode kernel of the previous batch, whose five waves per SIMD leave 32 VGPRs
// of each SIMD lane free (512 - 5 x 96).  A wave that fits there starts in a slot the decoder cannot use; one that
// does not waits for a decode workgroup to retire and keeps the next one out.  Hence the ballots live in LDS and the
// second phase is a rolled loop with one instance of the polar arithmetic.
__global__ __launch_bounds__(kScanThreads) void polar_compact_kernel(const uint64_t *raw, uint64_t n_trials,
                                                                     const uint64_t *block_offsets, uint64_t want,
                                                                     uint64_t *pairs_out, ScanResult *result)
{
    constexpr int kWaves = kScanThreads / 64;
    __shared__ unsigned long long ballot[kScanIters * kWaves]; // accepted lanes of (it, wave): trials in ascending order
    __shared__ int first[kScanIters * kWaves];                 // accepted trials of the block before (it, wave)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t off = block_offsets[blockIdx.x];
    if (off >= want) // nothing this block holds is needed
        return;
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanBlock;
#pragma unroll 2
    for (int it = 0; it < kScanIters; ++it)
    {
        const unsigned long long m = __ballot(trial_accepted(raw, base + it * kScanThreads + tid, n_trials));
        if (lane == 0)
            ballot[it * kWaves + wave] = m;
    }
    __syncthreads();
    if (tid == 0)
    {
        int run = 0;
        for (int i = 0; i < kScanIters * kWaves; ++i)
        {
            first[i] = run;
            run += __popcll(ballot[i]);
        }
    }
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int it = 0; it < kScanIters; ++it)
    {
        const unsigned long long m = ballot[it * kWaves + wave];
        if (!(m >> lane & 1))
            continue;
        const uint64_t rank = off + first[it * kWaves + wave] + __popcll(m & ((1ull << lane) - 1));
        if (rank >= want)
            continue;
        const uint64_t t = base + it * kScanThreads + tid;
        if (pairs_out)
        {
            // the accepted trial leaves here as its two normals (libstdc++ normal_distribution, polar method: y*mult is
            // returned first, x*mult saved for the next call), so the decode launch finds finished variates instead of
            // a log/divide/sqrt chain at the head of every frame
            const ulonglong2 u = *reinterpret_cast<const ulonglong2 *>(raw + 2 * t);
            const PolarTrial tr = polar_trial(u.x, u.y);
            __builtin_amdgcn_sched_barrier(0); // (stage by stage: the scheduler's interleaving costs 20 registers)
            const double lg = dm_log(tr.r2);
            __builtin_amdgcn_sched_barrier(0);
            const double q = -2 * lg / tr.r2;
            __builtin_amdgcn_sched_barrier(0);
            const double mult = __builtin_sqrt(q);
            __builtin_amdgcn_sched_barrier(0);
            ulonglong2 o;
            o.x = dm_bits(tr.y * mult), o.y = dm_bits(tr.x * mult);
            *reinterpret_cast<ulonglong2 *>(pairs_out + 2 * rank) = o;
        }
        if (rank == want - 1)
            result->trials_used = t + 1;
    }
}

} // namespace

int launch_mt_generate(const uint64_t *states, uint64_t *next_last, uint64_t *out, uint32_t n_chunks,
                       uint32_t chunk_words, int chunks_per_workgroup, void *stream)
{
    if (n_chunks == 0)
        return hipSuccess;
    if (chunk_words % kMtN != 0)
        return hipErrorInvalidValue;
    if (chunks_per_workgroup >= 4)
        hipLaunchKernelGGL(mt_generate_kernel<4>, dim3((n_chunks + 3) / 4), dim3(kGenThreads * 4), 0, static_cast<hipStream_t>(stream),
                           states, next_last, out, chunk_words, n_chunks);
    else
        hipLaunchKernelGGL(mt_generate_kernel<1>, dim3(n_chunks), dim3(kGenThreads), 0, static_cast<hipStream_t>(stream), states,
                           next_last, out, chunk_words, n_chunks);
    return hipGetLastError();
}

int launch_mt_jump(const uint64_t *src_states, uint64_t *dst_states, const uint64_t *poly, uint32_t n_tasks,
                   void *stream)
{
    if (n_tasks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(mt_jump_kernel, dim3(n_tasks), dim3(kJumpThreads), 0, static_cast<hipStream_t>(stream),
                       src_states, dst_states, poly);
    return hipGetLastError();
}

int launch_polar_scan(const uint64_t *raw, uint64_t n_trials, uint64_t want_pairs, uint32_t *block_counts,
                      uint64_t *block_offsets, uint64_t *pairs_out, ScanResult *result, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n_blocks = static_cast<uint32_t>((n_trials + kScanBlock - 1) / kScanBlock);
    if (n_blocks == 0)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(polar_count_kernel, dim3(n_blocks), dim3(kScanThreads), 0, s, raw, n_trials, block_counts);
    hipLaunchKernelGGL(polar_offsets_kernel, dim3(1), dim3(1024), 0, s, block_counts, n_blocks, block_offsets,
                       want_pairs, result);
    hipLaunchKernelGGL(polar_compact_kernel, dim3(n_blocks), dim3(kScanThreads), 0, s, raw, n_trials, block_offsets,
                       want_pairs, pairs_out, result);
    return hipGetLastError();
}

int launch_polar_count(const uint64_t *raw, uint64_t n_trials, uint64_t piece_trials, uint32_t *block_counts,
                       uint64_t *block_offsets, ScanResult *result, ScanResult *result_all, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n_blocks = static_cast<uint32_t>((n_trials + kScanBlock - 1) / kScanBlock);
    if (n_blocks == 0 || piece_trials % kScanBlock != 0 || piece_trials > n_trials)
        return hipErrorInvalidValue;
    const uint32_t piece_blocks = static_cast<uint32_t>(piece_trials / kScanBlock);
    hipLaunchKernelGGL(polar_count_kernel, dim3(n_blocks), dim3(kScanThreads), 0, s, raw, n_trials, block_counts);
    // offsets of the piece's blocks (gives the piece's total), then of all blocks (the ones the compaction uses)
    hipLaunchKernelGGL(polar_offsets_kernel, dim3(1), dim3(1024), 0, s, block_counts, piece_blocks, block_offsets, 0ull, result);
    hipLaunchKernelGGL(polar_offsets_kernel, dim3(1), dim3(1024), 0, s, block_counts, n_blocks, block_offsets, 0ull, result_all);
    return hipGetLastError();
}

int launch_polar_compact(const uint64_t *raw, uint64_t n_trials, const uint64_t *block_offsets, uint64_t want_pairs,
                         uint64_t *pairs_out, ScanResult *result, void *stream)
{
    const uint32_t n_blocks = static_cast<uint32_t>((n_trials + kScanBlock - 1) / kScanBlock);
    if (n_blocks == 0)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(polar_compact_kernel, dim3(n_blocks), dim3(kScanThreads), 0, static_cast<hipStream_t>(stream), raw, n_trials,
                       block_offsets, want_pairs, pairs_out, result);
    return hipGetLastError();
}

} // namespace ldpc_amd
