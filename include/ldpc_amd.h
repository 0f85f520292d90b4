/*
 * ldpc_amd.h — C ABI of libldpc.so, the MI355X-native drop-in for heat1q/libldpc's shared library.
 *
 * Part 1 re-exports, symbol for symbol and struct for struct, the six entry points of the
 * reference's src/shared.cpp:9-78 that pyLDPC/ldpc.py binds through ctypes, so an existing
 * `LDPC(pc_file, gen_file, lib="…/libldpc.so")` keeps working unchanged.
 * Part 2 is the batch interface the HIP path sits behind: the reference's per-frame virtual calls
 * (src/sim/channel.h:17-24 invoked at src/sim/ldpcsim.cpp:158-174) become one call per batch of
 * frames.  Plain pointers and sizes only; device pointers are accepted wherever a buffer is named
 * "device or host".
 */
#ifndef LDPC_AMD_H
#define LDPC_AMD_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ------------------------------------------------------------------------------------------ */
/* Part 1 — reference C ABI (src/core/functions.h:107-127, src/sim/ldpcsim.h:23-31)            */
/* ------------------------------------------------------------------------------------------ */
typedef struct
{
    bool earlyTerm;
    uint32_t iterations;
    const char *type; /* "BP_MS" selects min-sum, anything else sum-product (decoder.h:73-80) */
} decoder_param;

typedef struct
{
    uint64_t seed;
    double xRange[3]; /* MIN, MAX (exclusive), STEP */
    const char *type; /* "AWGN" | "BSC" | "BEC" */
} channel_param;

typedef struct
{
    uint32_t threads;  /* accepted for compatibility; the GPU batch replaces OpenMP threads */
    uint64_t maxFrames;
    uint64_t fec;
    const char *resultFile;
} simulation_param;

typedef struct
{
    double *fer;
    double *ber;
    double *avg_iter;
    double *time;
    uint64_t *fec;
    uint64_t *frames;
} sim_results_t;

/* replaces shared.cpp:11-24 — (re)creates the process-global code; file errors print and exit(1) */
void ldpc_setup(const char *pcFile, const char *genFile, int *n, int *m, int *nct, int *mct);
/* replaces shared.cpp:26-30 — blocks until done; writes results->x[i] per channel point; polls *stopFlag */
void simulate(decoder_param decoderParams, channel_param channelParam, simulation_param simParam,
              sim_results_t *results, bool *stopFlag);
/* replaces shared.cpp:32-35 */
int calculate_rank(void);
/* replaces shared.cpp:37-45 — infoWord[kct] -> codeWord[nct] (transmitted positions) */
void encode(uint8_t *infoWord, uint8_t *codeWord);
/* replaces shared.cpp:47-65 — llr[nct] in, llrOut[nct] out, returns the iteration count */
int decode(decoder_param decoderParams, double *llr, double *llrOut);
/* replaces shared.cpp:67-77 — word[nc] -> syndrome[mc] */
void syndrome(uint8_t *word, uint8_t *syndrome);

/* ------------------------------------------------------------------------------------------ */
/* Part 2 — batch interface (HIP shim)                                                         */
/* ------------------------------------------------------------------------------------------ */
typedef struct ldpc_hip_ctx ldpc_hip_ctx;

enum
{
    LDPC_HIP_AWGN = 1, /* ldpcsim.h:15-20 */
    LDPC_HIP_BSC = 2,
    LDPC_HIP_BEC = 3
};

/* outputs of a batch of n frames; each pointer is device or host memory, NULL = not wanted */
typedef struct
{
    uint32_t *iters;      /* [n]      iteration count as returned by ldpc_decoder::decode (decoder.cpp:74-77) */
    uint32_t *bit_errors; /* [n]      ldpcsim.cpp:184-188, transmitted positions only */
    uint8_t *hard;        /* [n][nc]  estimate() */
    double *llr_out;      /* [n][nc]  llr_out() (BEC: symbol values 0,1,'E') */
    double *llr_in;       /* [n][nc]  the decoder input the channel produced */
    uint8_t *codeword;    /* [n][nc]  transmitted codeword */
} ldpc_hip_out;

/* number of visible GPUs (0 when none); never throws */
int ldpc_hip_device_count(void);
/* message of the last failed ldpc_hip_* call on this thread */
const char *ldpc_hip_last_error(void);

/* parse the code (host only; the GPU is touched lazily by the first decode). NULL on error. */
ldpc_hip_ctx *ldpc_hip_create(const char *pcFile, const char *genFile, int device);
void ldpc_hip_destroy(ldpc_hip_ctx *ctx);
/* info[0..9] = nc, mc, nnz, nct, mct, kct, kc, max_degree, residency (0 memory, 1 LDS, 2 registers with a message
   mailbox, 3 registers with variable-node totals returned), lds_bytes_per_frame */
void ldpc_hip_code_info(const ldpc_hip_ctx *ctx, int64_t info[10]);
/* the text block the reference CLI prints for a code (ldpc.cpp:111-130); owned by the context */
const char *ldpc_hip_describe(ldpc_hip_ctx *ctx);
/* BEC: 1 = reproduce the reference's out-of-bounds read for erased degree-1 variable nodes
   (they emit 0); 0 (default) = defined semantics, they emit an erasure */
void ldpc_hip_set_bec_compat(ldpc_hip_ctx *ctx, int compat);
/* NON-PARITY modes for sum-product decoding ("BP"), off (0) by default and never chosen by the library itself; min-sum and
   BEC ignore them.  Error rates differ from the reference's within what profiles/ reports; iteration counts and decisions
   are not the reference's.
     1  flooding schedule, binary32 messages, LLRs clipped to +-27.7, hardware reciprocal / log2 / exp2
        (libldpc_amd/csrc/kernels_fast.hip)
     2  LAYERED (row-serial) schedule, binary32 check-to-variable messages: one wavefront per frame, a sweep is a sequence
        of conflict-free steps of up to 64 check nodes (libldpc_amd/csrc/kernels_layered.hip)
     3  the same with binary16 check-to-variable messages
   The reference has no such modes in src/ (its legacy gpu/ simulator uses single precision and processes H in layers,
   gpu/ldpc/ldpc.h, gpu/ldpc/ldpc.cpp:111-138: ideas only). */
void ldpc_hip_set_fast_mode(ldpc_hip_ctx *ctx, int mode);

/* decode n frames of given LLRs llr_in[n][nc] (column order, device or host). 0 on success. */
int ldpc_hip_decode_batch(ldpc_hip_ctx *ctx, decoder_param dec, uint64_t n, const double *llr_in,
                          const ldpc_hip_out *out, void *hip_stream);

/* channel point x of stream mt19937_64(seed): set_channel_param semantics, frame position := 0 */
int ldpc_hip_stream_begin(ldpc_hip_ctx *ctx, int channel, uint64_t seed, double x);
/* advance the stream by n frames without decoding them */
int ldpc_hip_stream_skip(ldpc_hip_ctx *ctx, uint64_t n, void *hip_stream);
/* channel + LLR init + decode of the next n frames of the stream, fused in one launch */
int ldpc_hip_stream_decode(ldpc_hip_ctx *ctx, decoder_param dec, uint64_t n, const ldpc_hip_out *out,
                           void *hip_stream);
/* the counters ldpc_sim::start accumulates per frame (ldpcsim.cpp:175-200), summed over a batch on the device:
   counters[0..4] = {frames, frame errors (bit_errors > 0), bit errors, iterations, frames that stopped early
   (iters < max_iters; 0 unless early_term)}.  iters / bit_errors / counters are DEVICE pointers (the outputs of a
   decode call on the same stream); one launch, ordered on hip_stream.  A multi-GPU run all-reduces these five. */
int ldpc_hip_batch_counters(ldpc_hip_ctx *ctx, const uint32_t *iters, const uint32_t *bit_errors, uint64_t n,
                            uint32_t max_iters, int early_term, int64_t *counters, void *hip_stream);
/* frames consumed / raw 64-bit draws consumed since ldpc_hip_stream_begin */
uint64_t ldpc_hip_stream_frame(const ldpc_hip_ctx *ctx);
uint64_t ldpc_hip_stream_raw_draws(const ldpc_hip_ctx *ctx);
int ldpc_hip_synchronize(ldpc_hip_ctx *ctx, void *hip_stream);

/* time kernels with HIP events: which = 0 the decode launches (recorded on the launch stream), 1 the noise-stream
   refills (jump-ahead + generator + slab table, recorded on the library's internal stream); host wall-clock: which = 2 the
   time the calling thread spent inside the ranks' exchange (all-gather) of a sharded step, 3 the time it waited for the
   noise stream's result (both: mean milliseconds per step since the previous call).  Event pairs are queued
   per launch; ldpc_hip_last_ms waits for them and returns the MEAN duration in milliseconds of the launches since
   the previous call (so a caller that reads it once per launch sees that launch, and a caller that reads it after
   a loop does not serialise the overlap of the noise stream with the decode) */
void ldpc_hip_set_profiling(ldpc_hip_ctx *ctx, int on);
float ldpc_hip_last_ms(ldpc_hip_ctx *ctx, int which);

/* first n outputs of std::mt19937_64(seed) starting at output `first`, produced by the device
   generator (jump-ahead + parallel chunks); out is device or host memory */
int ldpc_hip_mt64(ldpc_hip_ctx *ctx, uint64_t seed, uint64_t first, uint64_t n, uint64_t *out, void *hip_stream);

/* self-test of the device arithmetic: n pseudo-random positive operand pairs (a, b) with a, b and a/b inside
   2^-+1000 go through the division sequence the likelihood-ratio kernels use (detmath.h, dm_ratio_div) and through
   the IEEE division; *mismatches receives the number of pairs whose quotients differ in any bit (expected: 0) */
int ldpc_hip_selftest_division(ldpc_hip_ctx *ctx, uint64_t n, uint64_t seed, uint64_t *mismatches);

/* the kernels' arithmetic, element by element, for tests that hold it against libm and extended-precision host values
   rather than against the same header compiled for the host: out[i] = fn(a[i] [, b[i]]) for the scalar functions
   (fn 0..9: dm_exp, dm_log, dm_boxplus, dm_ratio_div, dm_ratio_rho, dm_ratio_lambda, dm_e_combine, dm_exp_clamped,
   dm_boxplus_exp, dm_boxplus_log of libldpc_amd/csrc/detmath.h), out[i][0..D) = check-node update of the row
   a[i][0..D) for fn 10..14 (likelihood-ratio form, D = 3, 4, 5, 6, 8) and fn 15, 16 (LLR domain, D = 4, 6).
   a, b, out are HOST buffers of n (x D) doubles; b may be NULL for one-operand functions. */
int ldpc_hip_selftest_math(ldpc_hip_ctx *ctx, int fn, uint64_t n, const double *a, const double *b, double *out);

/* host-only self-tests of the noise stream's chunk-state bookkeeping (libldpc_amd/csrc/mtstates.hpp; no GPU needed): the
   planned operations are replayed on a symbolic table in which every row records which chunk's state it holds.
   chunk_table: n_requests requests of chunks_per_request consecutive chunks each, `gap` chunks apart (0 = one rank reading
   the stream front to back; > 0 = a rank of a sharded BSC / BEC stream), starting at first_chunk.  Returns the number of
   jump-ahead tasks issued in total, or UINT64_MAX if an operation read a row without a valid state or a request was left
   without its rows.
   shard_table: `steps` sharded AWGN steps of rank `rank` of `world` (piece_chunks chunks + the margin chunk per rank and
   step).  Returns the jump-ahead tasks of the steps AFTER the first (and their launches in *launches): piece_chunks + 1
   per step whatever the world size — or UINT64_MAX as above. */
uint64_t ldpc_hip_selftest_chunk_table(uint64_t first_chunk, uint64_t chunks_per_request, uint64_t n_requests, uint64_t gap);
uint64_t ldpc_hip_selftest_shard_table(int world, int rank, uint32_t piece_chunks, uint64_t steps, uint64_t *launches);
/* jump-ahead tasks (one task = one chunk start state advanced by one polynomial) this context's noise stream has launched */
uint64_t ldpc_hip_jump_tasks(const ldpc_hip_ctx *ctx);

/* the simulation loop of ldpc_sim::start (ldpcsim.cpp:97-263) on one context; totals[4*i..] =
   {frames, fec, bec, iters} per channel point.  Returns the number of channel points, <0 on error. */
int ldpc_hip_simulate(ldpc_hip_ctx *ctx, decoder_param dec, channel_param ch, simulation_param sim,
                      sim_results_t *results, uint64_t *totals, bool *stopFlag, int cli_output);

/* ------------------------------------------------------------------------------------------ */
/* Part 3 — several GPUs, one process per GPU                                     */
/* ------------------------------------------------------------------------------------------ */
/* The reference shares its counters between OpenMP threads (ldpcsim.cpp:175-252); here the ranks of a sharded
   simulation exchange them, and the accepted-pair counts that place each rank in the one noise stream, through a
   communicator: RCCL over xGMI (one rank per GPU), or a host shared-memory segment for rehearsals in which ranks
   share a GPU and for tests without one.  All payloads are a few 64-bit words per rank. */
typedef struct ldpc_hip_comm ldpc_hip_comm;

/* rank 0 obtains an RCCL unique id (ncclGetUniqueId: 128 bytes) and hands the bytes to every rank by whatever means
   the launcher has (torch.distributed in bench.py, pipes in `ldpcsim --devices`) */
int ldpc_hip_comm_unique_id(uint8_t id[128]);
/* RCCL communicator of `world` ranks; this rank uses GPU `device` (ncclCommInitRank: collective, blocks) */
ldpc_hip_comm *ldpc_hip_comm_create(int rank, int world, int device, const uint8_t id[128]);
/* host shared-memory communicator; `name` ("/something") is the same on every rank and unique to the job */
ldpc_hip_comm *ldpc_hip_comm_create_shm(int rank, int world, const char *name);
/* one process standing in for rank `rank` of `world`, no transport: every rank's slot of an all-gather is answered with the
   caller's own payload.  For cost probes of the sharded step on one GPU (tools/shard_probe.py), not for results. */
ldpc_hip_comm *ldpc_hip_comm_create_echo(int rank, int world);
void ldpc_hip_comm_destroy(ldpc_hip_comm *comm);
/* recv[q*bytes ..) = rank q's send[0 .. bytes): host buffers, bytes a multiple of 8 and at most 256.  Never waits without a
   bound: when a rank does not arrive within LDPC_AMD_COMM_TIMEOUT_S seconds (default 60; ncclCommInitRank in
   ldpc_hip_comm_create likewise) the call fails on the ranks that wait for it and the communicator is unusable from then on */
int ldpc_hip_comm_allgather(ldpc_hip_comm *comm, const void *send, void *recv, uint64_t bytes);
/* the plan of the fused form of the first ratio launch (libldpc_amd/csrc/fused_rule.h decides which codes
   take it): info = {the code qualifies, message slots, variable-node blocks per wave, leaf calls per wave, the small
   instantiation applies, check-node calls per wave + 1, the code has shortened bits, entries of the slot table} */
void ldpc_hip_fused_plan_info(const ldpc_hip_ctx *ctx, int64_t info[8]);
/* the steps of the layered schedule of the non-parity modes 2 / 3 (host only): step_of_row[mc] = the step each check node
   is processed in; returns the number of steps, -1 when the schedule does not take the code */
int ldpc_hip_selftest_layer_plan(ldpc_hip_ctx *ctx, int32_t *step_of_row);
/* the placement step of ldpc_hip_stream_decode_sharded by itself, without a GPU (tests): this rank reports {pairs in its
   piece, pairs including the margin, status}; after the all-gather over `comm` out = {first frame, frames of this rank,
   frames of the step, stream index of the piece's first pair, stream index of the first pair after the step}.  A non-zero
   status on any rank, a piece with more frames than cap, or a frame beyond a margin fails the call on every rank. */
int ldpc_hip_selftest_place(ldpc_hip_comm *comm, uint64_t nct, uint64_t pairs_before, uint64_t frame_pos, uint64_t cap,
                            uint64_t piece_pairs, uint64_t pairs_with_margin, uint64_t status, uint64_t out[5]);
/* host microseconds spent inside the communicator's all-gathers since the last reset: out = {calls, min, median, max} */
void ldpc_hip_comm_stats(ldpc_hip_comm *comm, double out[4], int reset);
/* "rccl 2.22.3", "shm" or "echo" (valid until the communicator is destroyed) */
const char *ldpc_hip_comm_describe(ldpc_hip_comm *comm);

/* frames the output buffers of ldpc_hip_stream_decode_sharded must hold for a step of target_frames frames: a bound (every
   trial of a piece accepted), not a statistical estimate */
uint64_t ldpc_hip_shard_capacity(const ldpc_hip_ctx *ctx, uint64_t target_frames, int world);
/* this rank's share of the next global step of about target_frames frames of the stream (all ranks call it with the same
   arguments).  AWGN: the step is a range of the raw mt19937_64 stream cut into `world` pieces of whole generator chunks;
   each rank generates its own piece only (its chunk start states are the previous step's advanced by one polynomial), one
   all-gather of the accepted-pair counts tells every rank where its piece starts in the pair sequence, and a frame belongs
   to the rank whose piece holds its first pair.  The step holds whatever frames its raw range holds (at least one chunk
   per rank).  BSC / BEC: even split.  A failure on any rank makes every rank return an error from the same call.
   step[0..3] = first frame and frame count of the global step, first frame and frame count of this rank. */
int ldpc_hip_stream_decode_sharded(ldpc_hip_ctx *ctx, ldpc_hip_comm *comm, decoder_param dec, uint64_t target_frames,
                                   const ldpc_hip_out *out, uint64_t step[4], void *hip_stream);
/* ldpc_hip_simulate over the ranks of `comm`: every rank returns the counters of the one-rank run with the same
   arguments (the stop rule of ldpcsim.cpp:255 is applied in stream order across the ranks); rank 0 prints and writes
   the result file */
int ldpc_hip_simulate_sharded(ldpc_hip_ctx *ctx, ldpc_hip_comm *comm, decoder_param dec, channel_param ch,
                              simulation_param sim, sim_results_t *results, uint64_t *totals, bool *stopFlag,
                              int cli_output);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* LDPC_AMD_H */
