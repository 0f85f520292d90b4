// rng_kernels.hip — the reference's noise source rebuilt for the GPU.
//
// The reference draws all channel noise from ONE sequential std::mt19937_64(seed) per thread
// (src/sim/channel.cpp:5-15,37-42), through libstdc++'s normal_distribution (Marsaglia polar
// method with rejection: a data-dependent number of draws per sample).  To keep frame f the f-th
// frame of that very stream while decoding tens of thousands of frames per launch, the stream is
// cut into chunks of a fixed number of 312-word twist blocks and produced in data-parallel steps:
//
//   mt_jump_kernel      start state of a chunk from the start state of an earlier one, by GF(2) jump-ahead
//                       (state_{n+J} = g_J(T) state_n; evaluated as a sliding XOR of sequence
//                       words selected by the coefficients of g_J = t^J mod charpoly),
//   mt_generate_kernel  three waves per chunk regenerate its 312-word state in LDS block by block and write the tempered
//                       64-bit outputs,
//   polar_slab_kernel   AWGN: reads the raw words once: polar acceptance test of every trial (raw words 2t, 2t+1), and
//                       the two normals of each accepted trial, compacted in stream order into the chunk's slab (the
//                       workgroups of a chunk chain their counts by decoupled look-back: one pass, no counting launch),
//   normals_finish      prefix sums of the chunks' accepted-pair counts: the table by which the decode kernels find
//                       normal g as element g&1 of pair g>>1 (device_channel.hpp),
//                       BSC / BEC / info words use the raw words themselves: one draw per bit.
//
// The arithmetic of the acceptance test is device_math.hpp::polar_trial.
#include <hip/hip_runtime.h>

#include <cstdlib>

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
// workgroup barriers.  Several chunks share a workgroup so that a batch's 123 chunks occupy 31 compute units for the
// length of the serial chain and not 82: the register-resident decode kernel owns a whole CU per frame, and every CU
// that hosts a generator wave is lost to it for that long.
// (kGenChunks = 1 when the decode kernel shares its CUs anyway: packed chunks lengthen the serial chain, which then
// outlasts the headline decode kernel it runs under)
constexpr int kGenThreads = 192;

__device__ __forceinline__ uint32_t ring_index(uint32_t rows, uint32_t first, uint32_t i)
{
    const uint32_t r = first + i; // first < rows, i < rows
    return r >= rows ? r - rows : r;
}

// 32-bit data-parallel-primitive move on a 64-bit value (two halves): lane i receives lane i+1 (lane 63: 0)
__device__ __forceinline__ uint64_t dpp_wave_up1(uint64_t x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(x >> 32), 0x130, 0xF, 0xF, true);
    return (static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo);
}

// Thread t < 156 of a chunk's three waves holds the sequence words L = x[312 b + t] and H = x[312 b + 156 + t] of the
// current block b IN REGISTERS.  The twist
//     x[k + 312] = x[k + 156] ^ f(x[k], x[k + 1])
// needs the right-hand neighbour's pair (L, H): one whole-wave lane shift, plus four edge values per block that cross the
// waves through LDS (wave 1 lane 0, wave 2 lane 0, and wave 0 lanes 0 and 1 for thread 155, whose neighbour "156" is
// (H_0, new L_0)), double-buffered by the block's parity: ONE workgroup barrier per block where the LDS-resident state of
// rounds 1-2 needed three.  The serial chain per chunk is what the side stream's duration is made of.
template <int kGenChunks>
__global__ __launch_bounds__(kGenThreads *kGenChunks) void mt_generate_kernel(const uint64_t *ring, uint32_t ring_rows, uint32_t first_row,
                                                                              uint64_t *out, uint32_t chunk_words, uint32_t last_words,
                                                                              uint32_t n_chunks)
{
    __shared__ uint64_t edge[kGenChunks][2][4][2]; // [chunk of the workgroup][parity][w0 l0, w0 l1, w1 l0, w2 l0][L, H]
#ifdef LDPC_AMD_GEN_PRIO
    __builtin_amdgcn_s_setprio(LDPC_AMD_GEN_PRIO); // (experiments: the generator's few waves beside the decode kernel's, which run at 3)
#endif
    const int t = threadIdx.x % kGenThreads, sub = threadIdx.x / kGenThreads;
    const int lane = t & 63, w = t >> 6;
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kGenChunks + sub;
    const bool live = c < n_chunks; // (the last workgroup may hold fewer chunks; its idle waves still meet the barriers)
    uint64_t L = 0, H = 0;
    if (live && t < 156)
    {
        const uint64_t *s = ring + static_cast<size_t>(ring_index(ring_rows, first_row, static_cast<uint32_t>(c))) * kMtN;
        L = s[t], H = s[t + 156];
    }
    uint64_t *o = out + c * chunk_words;
    const uint32_t blocks = chunk_words / kMtN;
    // the launch's last chunk may be a prefix (last_words <= chunk_words): its waves keep meeting the barriers
    const uint32_t my_blocks = c + 1 == n_chunks ? last_words / kMtN : blocks;
    const int my_slot = w == 0 ? lane : w + 1; // where lanes 0 (and lane 1 of wave 0) publish
    const bool publisher = lane == 0 || (w == 0 && lane == 1);
    if (publisher)
        edge[sub][0][my_slot][0] = L, edge[sub][0][my_slot][1] = H;
    __syncthreads();
    for (uint32_t b = 0; b < blocks; ++b)
    {
        const int par = b & 1;
        if (live && t < 156 && b < my_blocks)
        {
            o[b * kMtN + t] = mt_temper(L);
            o[b * kMtN + 156 + t] = mt_temper(H);
        }
        uint64_t L1 = dpp_wave_up1(L), H1 = dpp_wave_up1(H);
        if (lane == 63 && w < 2) // thread 64 (w + 1) sits in the next wave
            L1 = edge[sub][par][w + 2][0], H1 = edge[sub][par][w + 2][1];
        if (t == 155) // x[156] = H_0 and x[312] = the NEW word 0, formed here from L_0, L_1, H_0
        {
            const uint64_t l0 = edge[sub][par][0][0], h0 = edge[sub][par][0][1], l1 = edge[sub][par][1][0];
            L1 = h0;
            H1 = mt_twist(l0, l1, h0);
        }
        const uint64_t nL = mt_twist(L, L1, H);
        const uint64_t nH = mt_twist(H, H1, nL);
        L = nL, H = nH;
        if (publisher)
            edge[sub][par ^ 1][my_slot][0] = L, edge[sub][par ^ 1][my_slot][1] = H;
        __syncthreads();
    }
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
constexpr int kJumpThreads = 192;    // per thread group: thread i < 156 owns output words 2i and 2i+1
constexpr int kJumpStages = 64;      // 64 * 312 = 19968 >= 19937 taps
constexpr int kPolyWords = 320;      // 312 coefficient words + zero padding read by the last stage

// kGroups thread groups per task share a stage's taps (group q takes the eights q, q + kGroups, ...) and XOR their partial
// windows together at the end: the task is a chain of 64 stages whose length is one group's share of the taps, and a batch
// has only a couple of hundred tasks — the chain's latency, not the chip's throughput, is what a jump costs (round 4: one
// group of 192 threads took 0.7 ms per launch, which bounded the erasure channel's whole step once its decoder had become
// bit-sliced; four groups with branches take a fraction of that).  The eight coefficient bits of a step are wave-uniform:
// with kBranch the XORs of a zero coefficient are BRANCHED over (an empty asm in the body keeps the compiler from turning the
// branch back into sixteen selects per step, of which half do nothing: the polynomial's density is one half).
// Which form runs where is a measurement (same box, A/B): the short chain wins where the jump is on the critical path (the
// erasure channel: 0.70 -> 0.39 ms of noise chain per step) and beside the register-resident decoders (its workgroup holds
// a CU a tenth as long: config 4 4.26 -> 4.20 ms); beside the LDS-resident sum-product decoders, whose issue slots it
// shares, the slow trickle of ONE group of masked XORs disturbs least (headline step 2.71 ms; one group with branches 2.73,
// four groups 2.75) and stays.
// kPack tasks per workgroup (their waves share nothing but the barriers): beside a decode kernel whose workgroup owns a
// whole CU, every resident jump workgroup keeps a frame out for as long as it runs.
template <int kPack, int kGroups, bool kBranch>
__global__ __launch_bounds__(kJumpThreads *kGroups *kPack) void mt_jump_kernel(uint64_t *table, uint32_t ring_rows, uint32_t src_first,
                                                                               uint32_t dst_first, const uint64_t *poly, uint32_t n_tasks)
{
    constexpr int kTask = kJumpThreads * kGroups; // threads per task
    __shared__ uint64_t xs[kPack][kMtN];                                  // generator state = newest block
    __shared__ __attribute__((aligned(16))) uint64_t rings[kPack][2 * kMtN];
    __shared__ uint64_t g[kPolyWords];
    __shared__ uint64_t part[kPack][kGroups > 1 ? kGroups - 1 : 1][kMtN]; // partial windows of groups 1..
    const int tid = threadIdx.x % kTask, sub = threadIdx.x / kTask;
    const int q = __builtin_amdgcn_readfirstlane(tid / kJumpThreads), t = tid % kJumpThreads; // (192 = three whole waves)
    const uint32_t task = blockIdx.x * kPack + sub;
    const bool live = task < n_tasks;
    uint64_t *x = xs[sub], *ring = rings[sub];
    const uint64_t *src = table + static_cast<size_t>(ring_index(ring_rows, src_first, live ? task : 0)) * kMtN;
    uint64_t *dst = table + static_cast<size_t>(ring_index(ring_rows, dst_first, live ? task : 0)) * kMtN;
    for (int k = threadIdx.x; k < kPolyWords; k += kTask * kPack)
        g[k] = poly[k];
    for (int k = tid; k < kMtN; k += kTask)
    {
        const uint64_t v = src[k];
        x[k] = v;
        ring[k] = v;
    }
    __syncthreads();
    if (tid < 64)
        mt_regenerate(x, tid);
    __syncthreads();
    for (int k = tid; k < kMtN; k += kTask)
        ring[kMtN + k] = x[k];
    __syncthreads();

    uint64_t acc0 = 0, acc1 = 0;
    for (int b = 0; b < kJumpStages; ++b)
    {
        if (t < kMtN / 2)
        {
            const int k0 = b * kMtN;
            const uint64_t *rp = ring + 2 * t; // 16-byte aligned: the eight taps' words for both outputs are nine consecutive words
            for (int kk = 8 * q; kk < kMtN; kk += 8 * kGroups)
            {
                // eight coefficient bits starting at k0 + kk (wave-uniform)
                const int k = k0 + kk;
                const int w = k >> 6, sh = k & 63;
                uint64_t lo = g[w] >> sh;
                uint64_t hi = sh ? (g[w + 1] << (64 - sh)) : 0;
                uint32_t bits = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(lo | hi))) & 0xFFu;
                if (bits == 0)
                    continue;
                uint64_t v[10];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const ulonglong2 u = *reinterpret_cast<const ulonglong2 *>(rp + kk + 2 * j);
                    v[2 * j] = u.x, v[2 * j + 1] = u.y;
                }
                v[8] = rp[kk + 8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (bits >> j & 1)
                    {
                        acc0 ^= v[j], acc1 ^= v[j + 1];
                        if constexpr (kBranch)
                            asm volatile(""); // (a real branch: see above)
                    }
            }
        }
        __syncthreads();
        // advance: block b+1 becomes the low half, wave 0 produces block b+2
        if (tid < 64)
            mt_regenerate(x, tid);
        uint64_t up[(kMtN + kTask - 1) / kTask];
#pragma unroll
        for (int r = 0; r < (kMtN + kTask - 1) / kTask; ++r)
            up[r] = tid + r * kTask < kMtN ? ring[kMtN + tid + r * kTask] : 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < (kMtN + kTask - 1) / kTask; ++r)
            if (tid + r * kTask < kMtN)
                ring[tid + r * kTask] = up[r], ring[kMtN + tid + r * kTask] = x[tid + r * kTask];
        __syncthreads();
    }
    if constexpr (kGroups > 1)
    {
        if (q > 0 && t < kMtN / 2)
            part[sub][q - 1][2 * t] = acc0, part[sub][q - 1][2 * t + 1] = acc1;
        __syncthreads();
        if (q == 0 && t < kMtN / 2)
#pragma unroll
            for (int r = 0; r < kGroups - 1; ++r)
                acc0 ^= part[sub][r][2 * t], acc1 ^= part[sub][r][2 * t + 1];
    }
    if (live && q == 0 && t < kMtN / 2) // (the task's own source row was read into LDS at the start: in place is fine)
        dst[2 * t] = acc0, dst[2 * t + 1] = acc1;
}

// ---- AWGN: raw chunk -> slab of normals -------------------------------------------------------------------------
constexpr int kSlabThreads = 256;
constexpr int kSlabIters = kSlabBlock / kSlabThreads; // trials per thread
constexpr unsigned long long kLbPrefix = 1ull << 63, kLbAggregate = 1ull << 62, kLbValue = (1ull << 62) - 1;

// One workgroup = 2048 consecutive trials of one chunk.  Phase 1: the acceptance test of the thread's eight trials (kept
// in registers), ballots to LDS, the workgroup's count.  Between the phases the workgroups of a chunk chain their counts
// by decoupled look-back (each publishes its aggregate, adds up its predecessors' until it meets a published prefix,
// publishes its own prefix): a single pass over the raw words, no counting launch before the compaction.  Workgroups take
// their place in their chunk's line from a ticket counter, so a workgroup only ever waits for workgroups that started before it.
// Phase 2: every accepted trial is written to the chunk's slab at its rank in stream order, as its two normals.
__global__ __launch_bounds__(kSlabThreads) void polar_slab_kernel(NormalsArgs a, uint32_t trials_full, uint32_t trials_last,
                                                                  uint32_t blocks_per_chunk)
{
    constexpr int kWaves = kSlabThreads / 64;
    __shared__ unsigned long long ballot[kSlabIters * kWaves]; // accepted lanes of (it, wave): trials in ascending order
    __shared__ uint32_t first[kSlabIters * kWaves];            // accepted trials of the workgroup before (it, wave)
    __shared__ uint32_t s_ticket, s_base, s_count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long *lb = reinterpret_cast<unsigned long long *>(a.lookback);
    // the workgroup's chunk is fixed by its index; its place in the chunk's line is a ticket of that chunk's counter (one
    // counter per chunk: a single counter for all 21 000 workgroups of a batch serialises them on one address)
    const uint32_t chunk = blockIdx.x / blocks_per_chunk;
    // (the launch's last chunk may be a prefix: fewer trials, fewer workgroups in its line)
    const uint32_t trials = chunk + 1 == a.n_chunks ? trials_last : trials_full;
    const uint32_t my_blocks = (trials + kSlabBlock - 1) / kSlabBlock;
    if (blockIdx.x % blocks_per_chunk >= my_blocks)
        return;
    if (tid == 0)
        s_ticket = atomicAdd(reinterpret_cast<unsigned int *>(lb + static_cast<uint64_t>(a.n_chunks) * blocks_per_chunk + chunk), 1u);
    __syncthreads();
    const uint32_t blk = s_ticket;
    const uint64_t *raw = a.raw + static_cast<uint64_t>(chunk) * (2ull * kBlockTrials * a.blocks);
    const uint32_t base = blk * kSlabBlock;
    PolarTrial tr[kSlabIters];
    bool acc[kSlabIters];
#pragma unroll
    for (int it = 0; it < kSlabIters; ++it)
    {
        const uint32_t t = base + it * kSlabThreads + tid;
        ulonglong2 u = {0, 0};
        if (t < trials)
            u = *reinterpret_cast<const ulonglong2 *>(raw + 2 * static_cast<uint64_t>(t));
        tr[it] = polar_trial(u.x, u.y);
        acc[it] = t < trials && tr[it].accepted();
        const unsigned long long m = __ballot(acc[it]);
        if (lane == 0)
            ballot[it * kWaves + wave] = m;
    }
    __syncthreads();
    if (wave == 0)
    {
        // the workgroup's 32 ballots -> exclusive counts (one lane each, a shuffle scan) and the workgroup's total
        static_assert(kSlabIters * kWaves <= 64, "one lane per ballot");
        const uint32_t own = lane < kSlabIters * kWaves ? __popcll(ballot[lane]) : 0u;
        uint32_t incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += up;
        }
        if (lane < kSlabIters * kWaves)
            first[lane] = incl - own;
        const uint32_t run = __shfl(incl, 63, 64);
        // decoupled look-back, a wave at a time: lane l inspects the predecessor l places back; the nearest published
        // prefix ends the walk, aggregates on the way are added.  The count travels in the same word as the flag: relaxed
        // atomics at device scope are all the ordering this needs.
        unsigned long long excl = 0;
        unsigned long long *mine = lb + static_cast<uint64_t>(chunk) * blocks_per_chunk;
        if (blk > 0)
        {
            if (lane == 0)
                __hip_atomic_store(mine + blk, kLbAggregate | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t j = static_cast<int64_t>(blk) - 1;
            for (;;)
            {
                const int64_t idx = j - lane;
                // (before block 0 of the chunk: a prefix of zero)
                const unsigned long long v = idx >= 0 ? __hip_atomic_load(mine + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kLbPrefix;
                const unsigned long long pm = __ballot((v & kLbPrefix) != 0), nr = __ballot(v == 0);
                const int cut = pm ? __builtin_ctzll(pm) : 63; // lanes 0..cut are needed
                const unsigned long long need = cut == 63 ? ~0ull : (2ull << cut) - 1;
                if (nr & need)
                {
                    __builtin_amdgcn_s_sleep(2); // a predecessor with an earlier ticket has not published yet: it is running
                    continue;
                }
                unsigned long long c = lane <= cut ? (v & kLbValue) : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
                    c += __shfl_xor(c, o, 64);
                excl += c;
                if (pm)
                    break;
                j -= 64;
            }
        }
        if (lane == 0)
        {
            __hip_atomic_store(mine + blk, kLbPrefix | (excl + run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = static_cast<uint32_t>(excl);
            s_count = run;
            if (blk + 1 == my_blocks)
                a.counts[chunk] = static_cast<uint32_t>(excl + run);
        }
    }
    __syncthreads();
    const uint32_t wg_base = s_base, wg_last = s_base + s_count - 1;
    uint64_t *slab = a.slabs + static_cast<uint64_t>(chunk) * a.slab_words;
    const bool locating = chunk == a.locate_chunk;
    const unsigned long long below = (1ull << lane) - 1;
#pragma unroll
    for (int it = 0; it < kSlabIters; ++it)
    {
        if (!acc[it])
            continue;
        const uint32_t rank = wg_base + first[it * kWaves + wave] + __popcll(ballot[it * kWaves + wave] & below);
        if (a.write_normals)
        {
            // the accepted trial leaves here as its two normals (libstdc++ normal_distribution, polar method: y*mult is
            // returned first, x*mult saved for the next call), so the decode launch finds finished variates instead of
            // a log/divide/sqrt chain at the head of every frame
            const double lg = dm_log(tr[it].r2);
            const double q = -2 * lg / tr[it].r2;
            const double mult = __builtin_sqrt(q);
            ulonglong2 o;
            o.x = dm_bits(tr[it].y * mult), o.y = dm_bits(tr[it].x * mult);
            *reinterpret_cast<ulonglong2 *>(slab + 2 * static_cast<uint64_t>(rank)) = o;
        }
        // (the chunk's last accepted pair: the last one of the chunk's last workgroup that holds any — later workgroups
        // of the chunk overwrite the answer of earlier ones only if they hold an accepted pair themselves; stream order
        // of the writes does not matter because only the chunk's final count decides, see the host side)
        if (locating && (a.locate_rank == 0xFFFFFFFFu ? rank == wg_last : rank == a.locate_rank))
        {
            const uint64_t t = base + it * kSlabThreads + tid;
            if (a.locate_rank == 0xFFFFFFFFu)
                atomicMax(reinterpret_cast<unsigned long long *>(a.locate_out), static_cast<unsigned long long>(t));
            else
                *a.locate_out = t;
        }
    }
}

// exclusive prefix sums of the chunk counts + where the consumer stands afterwards (kernels.hpp, NormalsResult)
__global__ __launch_bounds__(256) void normals_finish_kernel(const uint32_t *counts, uint32_t n, uint32_t n_piece, uint32_t n_full,
                                                             uint64_t need, uint64_t target, uint64_t *cum, NormalsResult *result)
{
    __shared__ uint64_t part[256];
    const int tid = threadIdx.x;
    const uint32_t per = (n + 255) / 256;
    const uint32_t lo = min(tid * per, n), hi = min(lo + per, n);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i)
        s += counts[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) // Hillis-Steele inclusive scan
    {
        const uint64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint64_t runsum = tid ? part[tid - 1] : 0;
    const uint64_t total = part[255];
    if (tid < 4)
        cum[n + 1 + tid] = ~0ull; // (the decode kernels read four consecutive entries around a slab: no slab beyond the last)
    if (tid == 0)
    {
        cum[n] = total;
        result->total = total;
        result->enough = total >= need;
        if (target >= total) // beyond everything generated
        {
            const uint32_t j = n_full >= n ? n : n - 1;
            result->next_slab = j;
            result->next_k = target - (j == n ? total : total - counts[n - 1]);
        }
        if (n_piece >= n)
            result->piece = total;
    }
    for (uint32_t i = lo; i < hi; ++i)
    {
        cum[i] = runsum;
        if (i == n_piece)
            result->piece = runsum;
        const uint64_t next = runsum + counts[i];
        if (runsum <= target && target < next)
        {
            result->next_slab = i;
            result->next_k = target - runsum;
        }
        runsum = next;
    }
}

} // namespace

int launch_mt_generate(const uint64_t *ring, uint32_t ring_rows, uint32_t first_row, uint64_t *out, uint32_t n_chunks,
                       uint32_t chunk_words, uint32_t last_words, int pack, void *stream)
{
    if (n_chunks == 0)
        return hipSuccess;
    if (chunk_words % kMtN != 0 || last_words % kMtN != 0 || last_words > chunk_words || first_row >= ring_rows || n_chunks > ring_rows)
        return hipErrorInvalidValue;
    if (pack >= 4)
        hipLaunchKernelGGL(mt_generate_kernel<4>, dim3((n_chunks + 3) / 4), dim3(kGenThreads * 4), 0, static_cast<hipStream_t>(stream),
                           ring, ring_rows, first_row, out, chunk_words, last_words, n_chunks);
    else
        hipLaunchKernelGGL(mt_generate_kernel<1>, dim3(n_chunks), dim3(kGenThreads), 0, static_cast<hipStream_t>(stream), ring,
                           ring_rows, first_row, out, chunk_words, last_words, n_chunks);
    return hipGetLastError();
}

int launch_mt_jump(uint64_t *ring, uint32_t ring_rows, uint32_t src_first, uint32_t dst_first, const uint64_t *poly,
                   uint32_t n_tasks, int pack, int groups, void *stream)
{
    if (n_tasks == 0)
        return hipSuccess;
    if (src_first >= ring_rows || dst_first >= ring_rows || n_tasks > ring_rows)
        return hipErrorInvalidValue;
    // groups = 4: one task per workgroup in four groups with branches (a short chain); else the one-group kernel with masked
    // XORs, `pack` tasks to a workgroup (LDPC_AMD_JUMP_GROUPS: experiments)
    static const int groups_env = std::getenv("LDPC_AMD_JUMP_GROUPS") ? std::atoi(std::getenv("LDPC_AMD_JUMP_GROUPS")) : 0;
    if (groups_env)
        groups = groups_env;
    if (groups >= 4)
        hipLaunchKernelGGL((mt_jump_kernel<1, 4, true>), dim3(n_tasks), dim3(kJumpThreads * 4), 0, static_cast<hipStream_t>(stream), ring, ring_rows,
                           src_first, dst_first, poly, n_tasks);
    else if (pack >= 3)
        hipLaunchKernelGGL((mt_jump_kernel<3, 1, false>), dim3((n_tasks + 2) / 3), dim3(kJumpThreads * 3), 0, static_cast<hipStream_t>(stream), ring,
                           ring_rows, src_first, dst_first, poly, n_tasks);
    else
        hipLaunchKernelGGL((mt_jump_kernel<1, 1, false>), dim3(n_tasks), dim3(kJumpThreads), 0, static_cast<hipStream_t>(stream), ring, ring_rows,
                           src_first, dst_first, poly, n_tasks);
    return hipGetLastError();
}

int launch_mt_normals(const NormalsArgs &a, void *stream)
{
    if (a.n_chunks == 0 || a.blocks == 0 || a.last_blocks == 0)
        return hipSuccess;
    const uint32_t trials = kBlockTrials * a.blocks;
    if (a.first_row >= a.ring_rows || a.n_chunks > a.ring_rows || a.last_blocks > a.blocks || a.slab_words < 2ull * trials || !a.raw || !a.lookback)
        return hipErrorInvalidValue;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = launch_mt_generate(a.ring, a.ring_rows, a.first_row, a.raw, a.n_chunks, kMtN * a.blocks, kMtN * a.last_blocks, a.pack, stream);
    if (rc != hipSuccess)
        return rc;
    const uint32_t bpc = (trials + kSlabBlock - 1) / kSlabBlock;
    rc = hipMemsetAsync(a.lookback, 0, 8 * normals_lookback_words(a.n_chunks, a.blocks), s);
    if (rc != hipSuccess)
        return rc;
    hipLaunchKernelGGL(polar_slab_kernel, dim3(a.n_chunks * bpc), dim3(kSlabThreads), 0, s, a, trials, kBlockTrials * a.last_blocks, bpc);
    return hipGetLastError();
}

int launch_normals_finish(const uint32_t *counts, uint32_t n, uint32_t n_piece, uint32_t n_full, uint64_t need, uint64_t target,
                          uint64_t *cum, NormalsResult *result, void *stream)
{
    if (n == 0)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(normals_finish_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), counts, n, n_piece, n_full, need,
                       target, cum, result);
    return hipGetLastError();
}

} // namespace ldpc_amd
