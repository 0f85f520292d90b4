// kernels.hpp — host-visible launch interface of the HIP kernels (kernels.hip).
#pragma once

#include <cstdint>

#include "plan.hpp"

namespace ldpc_amd
{

// Device copy of Plan: all pointers are device pointers owned by the engine.
struct DevPlan
{
    int nc, mc, nnz, nct;
    int n_bitpos;
    int n_cn_blocks, n_vn_blocks;
    int cn_work_stride, vn_work_stride;
    const CnBlock *cn_blocks;
    const VnBlock *vn_blocks;
    const uint32_t *vn_slot;
    const uint32_t *cn_work;
    const uint32_t *vn_work;
    const CnBlock *cn_work_desc; // [kDecodeWaves][cn_desc_stride], count 0 = none
    int cn_desc_stride;
    const uint32_t *vn_work_desc; // [kDecodeWaves][vn_work_stride + 1][4], count 0 = none
    const uint32_t *vn_packed;    // [kDecodeWaves][16][64] register-held slot indices as the lanes keep them (plan.hpp), or null
    const uint32_t *col_rank;
    const uint32_t *rank_col;
    const uint32_t *tx_rank;
    const uint8_t *rank_kind;
    const uint32_t *rank_slot0;
    const int *bit_pos; // [n_bitpos] transmitted index -> column
    uint32_t lds_bytes;
};

enum ChannelMode : int
{
    kModeLlr = 0,  // LLRs given per frame (C-ABI decode, shared.cpp:47-65)
    kModeAwgn = 1, // fused channel_awgn::simulate + calculate_llrs (channel.cpp:62-93)
    kModeBsc = 2,  // fused channel_bsc (channel.cpp:129-162)
};

struct DecodeArgs
{
    DevPlan plan;
    uint32_t iterations;
    int early_term;
    int mode;
    uint64_t n_frames;
    // kModeLlr
    const double *llr_in; // [n_frames][nc], column order
    // kModeAwgn: the two normals (bit patterns of y*mult, x*mult) of every accepted polar pair of the stream, as the noise
    // generator left them: one SLAB per generator chunk, compacted inside the slab in stream order (rng_kernels.hip,
    // polar_slab_kernel).  slab_cum[j] = pairs in the slabs before j (slab_cum[0] = 0, n_slabs + 1 entries, followed by four
    // entries of 2^64-1); the pair with
    // stream index q sits in the slab j with slab_cum[j] <= q - pair_origin < slab_cum[j+1], at
    // pairs[j * slab_words + 2 * (q - pair_origin - slab_cum[j])].  Normal g = frame*nct + i comes from pair g>>1.
    const uint64_t *pairs;
    const uint64_t *slab_cum;
    uint64_t slab_words;      // 64-bit words per slab (2 x the pairs a chunk can hold)
    uint32_t n_slabs;
    float slab_pairs_inv;     // 1 / (expected pairs per slab): first guess of the slab of a frame
    uint64_t pair_origin;     // stream index of the first pair of slab 0
    uint64_t normal_base;     // index of this batch's first normal in the stream
    int pairs_buffer;         // host bookkeeping: which of the engine's two slab buffers `pairs` points into
    double sigma, sigma2; // sqrt(sigma2), sigma2 = 10^(-snr/10)
    double inv_sigma2;    // 1 / sigma2, correctly rounded (host division): detmath.h, dm_div_by
    double shorten_llr;   // 99999.9 (AWGN) or delta (BSC)
    // kModeBsc: raw draws, one per transmitted bit: raw[frame*nct + i]
    const uint64_t *raw;
    double eps, delta;
    // transmitted codeword per frame [n_frames][nc] (nullptr = all-zero codeword)
    const uint8_t *codeword;
    // outputs (any may be nullptr)
    uint32_t *iters;
    uint32_t *bit_errors;
    uint8_t *hard;        // [n_frames][nc]
    double *llr_out;      // [n_frames][nc]
    double *llr_in_dump;  // [n_frames][nc]
    // memory-resident decoder only: per-frame state in device memory instead of LDS
    double *ws_msg;  // [n_frames][nnz]
    double *ws_llr;  // [n_frames][nc]
    uint8_t *ws_hb;  // [n_frames][nnz]
    double *ws_scr;  // [n_frames][2 nnz] scratch of check nodes wider than 16 (nullptr when the code has none);
                     // totals-form register kernel: [n_frames][(nv0 + nv1) * nt] channel terms of the variable nodes
    // sum-product in likelihood-ratio form (detmath.h): when redo_list is set the launch runs that form and
    // appends the frames it could not finish to redo_list[atomicAdd(redo_count)]; a launch with redo_list_in /
    // redo_count_in set decodes exactly those frames (block b takes frame redo_list_in[b], b < *redo_count_in)
    uint32_t *redo_list;
    uint32_t *redo_count;
    int ratio_separate; // ratio form with every check-node output divided separately (the middle launch of three, LDS-resident)
    const uint32_t *redo_list_in;
    const uint32_t *redo_count_in;
    // hand-over (sum-product without early termination, detmath.h "Hand-over"): redo_iter[pos] = iteration the frame
    // redo_list[pos] resumes at in the LLR domain (0xFFFFFFFF: decode it from scratch), ws_handover[pos][nnz] = its c2v
    // messages as the ratio form left them (lambda, a decision in the sign bit), in message-slot order
    uint32_t *redo_iter;
    const uint32_t *redo_iter_in;
    double *ws_handover;
    int handover_llr; // ws_handover holds the c2v messages as LLRs already (the fused form's hand-over), not as lambda
    uint64_t *phase_trace; // debug builds with -DLDPC_AMD_PHASE_TRACE only: [2048 frames of mid-launch][4 waves][8 values]
};

// device copy of RegPlan (register-resident decoder, kernels_reg.hip)
struct DevRegPlan
{
    int nt, kc, maxd, rounds;
    uint32_t mb_doubles;
    const uint32_t *cn_edge;
    const uint8_t *cn_deg;
    const uint8_t *cn_cnt;
    const RegVnBlock *vn_blocks;
    const uint32_t *round_first;
};

// device copy of Reg2Plan (register-resident decoder, second form: kernels_reg2.hip)
struct DevReg2Plan
{
    int nt, kc, maxd, nv0, nv1;
    uint32_t neutral, lds_entries;
    int uniform_cn; // every check-node block has maxd edges
    int uniform_vn; // every variable-node block is full, of degree 3, at the affine offsets below
    // round r: column position 0 of block q = i * (nt/64) + wave at entry vn_affine[r][0] + vn_affine[r][1] * q (+ lane),
    // positions 1.. at [2] + [3] * q (+ 64 per position), the total at [4] + [5] * q
    uint32_t vn_affine[2][6];
    const uint32_t *edge_w;
    const uint8_t *cn_deg;
    const Reg2VnBlock *vn_blocks;
    const uint32_t *vn_rank;
};

// device copy of FusedPlan (fused form of the likelihood-ratio iteration: kernels_fused.hip)
struct DevFusedPlan
{
    int n_slots, vnb, cnl, calls_stride;
    int has_shortened;
    int need_lambda; // a column that is transmitted or shortened has three or more edges: the prologue stages lambda_ch too
    int wide_exclusive;
    uint32_t vn_prog[kDecodeWaves]; // plan.hpp, FusedPlan::vn_prog
    uint32_t lds_bytes;          // dynamic LDS per frame: the message slots, or the staging area of the prologue if larger
    const FusedCall *leaf_calls; // [kDecodeWaves][kFusedLeafCalls]
    const FusedCall *calls;      // [kDecodeWaves][calls_stride]
    const uint32_t *vn_desc;     // [kDecodeWaves][kFusedVnSlots][4]
    const uint32_t *vn_slot;
    const uint32_t *lane_tab;    // [kDecodeWaves][kFusedLaneRows][64]
    const uint32_t *ho_map;      // [n_slots] (plan.hpp, FusedPlan::ho_map)
};
// first launch of sum-product with early termination (a.redo_list / a.redo_count set, a.early_term, no a.redo_count_in)
int launch_decode_fused(const DecodeArgs &a, const DevFusedPlan &f, void *stream);
// the same form WITHOUT early termination: separately divided outputs, hand-over to the LLR-domain form (a.redo_iter,
// a.ws_handover as for the general kernel; the messages arrive there as LLRs: set handover_llr for the resuming launch)
int launch_decode_fused_handover(const DecodeArgs &a, const DevFusedPlan &f, void *stream);
// min-sum without early termination on the same plan (no redo lists)
int launch_decode_fused_minsum(const DecodeArgs &a, const DevFusedPlan &f, void *stream);

// BEC (u8 erasure alphabet, decoder.cpp:91-192 + channel.cpp:199-229)
struct BecArgs
{
    DevPlan plan;
    uint32_t iterations;
    int early_term;
    int deg1_compat; // 1: erased degree-1 VN emits 0 (what the reference's out-of-bounds read yields)
    uint64_t n_frames;
    const uint64_t *raw; // raw[frame*nct + i], nullptr when symbols are given
    double eps;
    const uint8_t *symbols;  // [n_frames][nc] channel LLR alphabet {0,1,'E'} (when raw == nullptr)
    const uint8_t *codeword; // [n_frames][nc] or nullptr
    uint32_t *iters;
    uint32_t *bit_errors;
    uint8_t *hard;
    double *llr_out;     // symbol values 0, 1, 'E' widened to double
    double *llr_in_dump;
    uint8_t *ws;         // per-frame state in device memory (codes whose nnz + 2 nc bytes exceed LDS), else nullptr
};

// erasure decoder, 64 frames per workgroup (kernels_bec.hip); launch_bec (kernels.hip) takes it when the code fits
bool bec_sliced_fits(const DevPlan &p);
int launch_bec_sliced(const BecArgs &a, void *stream);

// All launchers enqueue on `stream` (hipStream_t passed as void*) and return a hipError_t as int.
// LDS-resident decoder; llr_mode: 0 input LLRs in LDS, 1 in device memory (a.ws_llr), 2 in registers
// (needs plan.vn_work_stride <= 8 and no isolated variable node)
int launch_decode_lds(const DecodeArgs &a, bool min_sum, int max_cn_degree, int llr_mode, void *stream);
// memory-resident variant for codes whose messages do not fit LDS (a.ws_* must be set); occupancy_lds
// bytes of dynamic LDS are requested only to bound the number of resident frames per CU
int launch_decode_mem(const DecodeArgs &a, bool min_sum, int max_cn_degree, uint32_t occupancy_lds, void *stream);
// register-resident decoder: a.ws_llr [n][nc] doubles and a.ws_hb [n][nc] bytes must be set
int launch_decode_reg(const DecodeArgs &a, const DevRegPlan &r, bool min_sum, void *stream);
// register-resident decoder, second form (same workspace requirements)
int launch_decode_reg2(const DecodeArgs &a, const DevReg2Plan &r, bool min_sum, void *stream);
// opt-in non-parity fast mode (kernels_fast.hip): sum-product with binary32 messages; a.ws_hb [n][nc] must be set
bool fast_mode_supported(const DevPlan &p, int max_cn_degree);
int launch_decode_fast(const DecodeArgs &a, int max_cn_degree, void *stream);
// opt-in non-parity layered schedule (kernels_layered.hip): device copy of LayerPlan (plan.hpp)
struct DevLayerPlan
{
    const uint32_t *steps; // [n_steps][2]: offset into vn / the message array, count | degree << 16
    const uint32_t *vn4;   // [(step * 4 + w) * 64 + lane]: VN ranks of edges 2w, 2w+1 of the step's lane-th check node (16 bits each)
    uint32_t n_steps, slots;
    uint32_t region_bytes, region_bytes_half; // LDS per frame with binary32 / binary16 messages
};
int launch_decode_layered(const DecodeArgs &a, const DevLayerPlan &L, bool half_messages, void *stream);
int launch_bec(const BecArgs &a, void *stream);

// ---- mt19937_64 on the device ----
constexpr int kMtN = 312;
constexpr uint32_t kBlockTrials = kMtN / 2; // one twist block of 312 words = 156 polar trials
// Chunk start states live in a RING of ring_rows rows of 312 words: chunk i of a launch starts from row
// (first_row + i) % ring_rows (first_row < ring_rows, n <= ring_rows).
// Raw (tempered) outputs: chunk i writes out[i * chunk_words ..), chunk_words a multiple of 312; the LAST chunk of the
// launch generates last_words <= chunk_words only (a prefix, in the same launch).  pack: chunks per workgroup (1, or 4 to
// keep the generator on a quarter of the compute units beside decode kernels that own a whole CU).
int launch_mt_generate(const uint64_t *ring, uint32_t ring_rows, uint32_t first_row, uint64_t *out, uint32_t n_chunks,
                       uint32_t chunk_words, uint32_t last_words, int pack, void *stream);
// Jump: row (dst_first + t) % ring_rows = row (src_first + t) % ring_rows advanced by the polynomial `poly` (19937
// coefficient bits, kJumpPolyWords words: 312 + zero padding), t < n_tasks.  A task reads its source row before it writes,
// so src_first == dst_first (in place) is allowed; otherwise the two row ranges must not overlap.
// groups: 4 = one task per workgroup, its taps shared by four thread groups (a short chain: where the jump is on the critical
// path or holds a CU a decoder wants); else one group, `pack` tasks per workgroup (1 or 3) — rng_kernels.hip says which where.
int launch_mt_jump(uint64_t *ring, uint32_t ring_rows, uint32_t src_first, uint32_t dst_first, const uint64_t *poly,
                   uint32_t n_tasks, int pack, int groups, void *stream);
constexpr uint32_t kJumpPolyWords = 320;

// The AWGN noise generator: chunks of the raw stream -> the normal variates the reference's normal_distribution would
// return (libstdc++ polar method).  Two launches: mt_generate writes the chunks' raw words, polar_slab_kernel
// reads them ONCE — acceptance test of every trial (raw words 2t, 2t+1), and for each accepted trial its two normals
// y*mult, x*mult, compacted in stream order into the chunk's SLAB (the workgroups of a chunk, 2048 trials each, chain
// their counts by decoupled look-back: no separate counting pass, no global prefix sum between the test and the
// compaction).  counts[i] = accepted trials of chunk i.
struct NormalsArgs
{
    const uint64_t *ring;
    uint32_t ring_rows, first_row;
    uint32_t n_chunks;
    uint32_t blocks;      // twist blocks (312 words = 156 trials each) a chunk holds
    uint32_t last_blocks; // ... and how many of them the launch's LAST chunk generates (a prefix; == blocks: all of it)
    int pack;             // generator: chunks per workgroup (launch_mt_generate)
    uint64_t *raw;        // scratch: n_chunks * 312 * blocks words
    uint64_t *lookback;   // scratch: n_chunks * (ceil(156 * blocks / kSlabBlock) + 1) words (zeroed by the launcher)
    uint64_t *slabs;      // chunk i writes slabs[i * slab_words ..): 2 words per accepted pair
    uint64_t slab_words;
    uint32_t *counts;     // [n_chunks]
    int write_normals;    // 0: count only (stream_skip, locate)
    // locate: in chunk locate_chunk (index in this launch; 0xFFFFFFFF = none) the chunk-local trial index of the accepted
    // pair with chunk-local rank locate_rank (0xFFFFFFFF = the chunk's last accepted pair) is written to *locate_out
    uint32_t locate_chunk, locate_rank;
    uint64_t *locate_out;
};
constexpr uint32_t kSlabBlock = 2048; // trials per workgroup of the slab kernel
inline uint64_t normals_lookback_words(uint32_t n_chunks, uint32_t blocks)
{
    return static_cast<uint64_t>(n_chunks) * ((static_cast<uint64_t>(kBlockTrials) * blocks + kSlabBlock - 1) / kSlabBlock + 1);
}
int launch_mt_normals(const NormalsArgs &a, void *stream);

// One small launch after the generator: cum[0..n] = exclusive prefix sums of counts[0..n) (the decode kernels' slab
// table), and where the consumer stands afterwards.
struct NormalsResult
{
    uint64_t total;      // cum[n]
    uint64_t piece;      // cum[n_piece] (sharded stream: accepted pairs of the rank's piece, without the margin chunk)
    uint64_t next_k;     // ... at this chunk-local pair index
    uint32_t next_slab;  // the pair with list index `target` sits in this slab ...
    uint32_t enough;     // total >= need
};
// target: list index (over the concatenated slabs) of the first pair the consumer has NOT used; n_full: leading chunks that
// were generated to full length (n or n - 1).  When target lies beyond everything generated, the position is
// (n, target - total) if the last chunk is complete and (n - 1, target - cum[n-1]) if it is a prefix that can be extended.
int launch_normals_finish(const uint32_t *counts, uint32_t n, uint32_t n_piece, uint32_t n_full, uint64_t need, uint64_t target,
                          uint64_t *cum, NormalsResult *result, void *stream);

// GF(2) encoding on the reference's info-word stream (channel.cpp:44-60, sparse.h:163-172).
// The reference draws kc bernoulli(0.5) bits per frame from mt19937_64(seed << 1) and ACCUMULATES u*G
// into the codeword it never clears, so the codeword of frame f is cw_prev ^ (u_0 ^ ... ^ u_f) G:
//   encode_info_kernel   info bit i of frame f = canonical(info_raw[f*kc + i]) < 0.5, packed 64 per word
//   encode_prefix_kernel running XOR of the packed info words over the frames of the batch
//   encode_cw_kernel     codeword[f][j] = cw_prev[j] ^ parity of (prefix_f AND column j of G)
struct EncodeArgs
{
    int nc, kc, words;         // words = ceil(kc/64)
    const uint32_t *g_col_ptr; // [g_cols+1] CSC of G (row indices per column)
    const uint32_t *g_col_row; // [g_nnz]
    int g_cols;
    const uint64_t *info_raw;  // [n_frames][kc]
    uint64_t *prefix;          // [n_frames][words] scratch / result
    const uint8_t *cw_prev;    // [nc]
    uint8_t *codeword;         // [n_frames][nc], may be nullptr (skip: only cw_last is wanted)
    uint8_t *cw_last;          // [nc] codeword after the batch
    uint64_t n_frames;
    // sharded encoding: XOR of the info words of the step's frames before this rank's (0 for a whole stream): joins every
    // prefix of the batch (the codeword accumulates linearly over the frames: channel.cpp:44-60, sparse.h:163-172)
    const uint64_t *base;      // [words] device, or nullptr
    // the columns of G as bit masks over the info word, [nc][words] (zero rows beyond g_cols), or nullptr: with them and
    // words <= 4 a codeword bit is a handful of AND / popcount instead of a walk over the column's entries
    const uint64_t *g_mask;
};
int launch_encode(const EncodeArgs &a, void *stream);
// the two halves of launch_encode by themselves (sharded encoding): info words + their running XOR over the batch; the
// codewords of the batch (only_last: just cw_last, from the batch's last prefix)
int launch_encode_prefix(const EncodeArgs &a, void *stream);
int launch_encode_codewords(const EncodeArgs &a, void *stream, bool only_last);

// {frames, frame errors, bit errors, iterations, early stops} of a batch (all device pointers), one launch
int launch_batch_counters(const uint32_t *iters, const uint32_t *bit_errors, uint64_t n, uint32_t max_iters, int early_term,
                          long long *counters, void *stream);

// dm_ratio_div vs the IEEE division on n pseudo-random operand pairs; *mismatches (device, zeroed by the caller)
int launch_division_selftest(uint64_t n, uint64_t seed, unsigned long long *mismatches, void *stream);

// functions of detmath.h / device_cn.hpp evaluated element by element (include/ldpc_amd.h: ldpc_hip_selftest_math)
enum MathFn : int
{
    kMathExp = 0,       // dm_exp(a)
    kMathLog,           // dm_log(a)
    kMathBoxplus,       // dm_boxplus(a, b)
    kMathRatioDiv,      // dm_ratio_div(a, b)
    kMathRatioRho,      // dm_ratio_rho(a, b)
    kMathRatioLambda,   // dm_ratio_lambda(a, b)
    kMathECombine,      // dm_e_combine(a, b)
    kMathExpClamped,    // dm_exp_clamped(a)
    kMathBoxplusExp,    // dm_boxplus_exp(a)
    kMathBoxplusLog,    // dm_boxplus_log(a)
    kMathCnRatio3,      // cn_ratio<3> on rows a[i][3] -> out[i][3]
    kMathCnRatio4,
    kMathCnRatio5,
    kMathCnRatio6,
    kMathCnRatio8,
    kMathCnLlr4,        // cn_core<4, sum-product> (LLR domain, E-form recursion)
    kMathCnLlr6,
    kMathCnRatio3s,     // dm_cn3_shared / dm_cn4_shared: the shared-reciprocal forms
    kMathCnRatio4s,
    kMathCnRatio6s,     // dm_cn6_shared: degree 6, two reciprocals
    kMathCount
};
int math_selftest_width(int fn); // values per element in a / out (0 = unknown function)
int launch_math_selftest(int fn, uint64_t n, const double *a, const double *b, double *out, void *stream);

} // namespace ldpc_amd
