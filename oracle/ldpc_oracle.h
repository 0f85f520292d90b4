/*
 * ldpc_oracle.h — CPU oracle for the LDPC belief-propagation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference algorithm
 * (heat1q/libldpc: src/decoding/decoder.{h,cpp}, src/sim/channel.cpp, src/sim/ldpcsim.cpp,
 * src/core/ldpc.cpp + sparse.h loaders) used as the checker by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing in the product
 * (libldpc_amd/, libldpc.so, ldpcsim) includes, links or calls it.
 *
 * Parity pin: the reference ships no decoder vectors, so this oracle is pinned
 * against the reference itself compiled from /root/reference by oracle/Makefile into
 * oracle/_ref/ (see oracle/ref_driver.cpp, tests/golden/make_golden.py and the committed
 * fixtures under tests/golden/).  With ORC_MATH_LIBM it is bit-identical to the reference
 * on this toolchain (glibc 2.35 libm, libstdc++ 11 <random> semantics restated below).
 * With ORC_MATH_DET the transcendental arithmetic is that of libldpc_amd/csrc/detmath.h — the same routines
 * and the same algebraic forms the HIP kernels use (deterministic exp/log; the check node in E = e^-|L|; for
 * sum-product with early termination the whole iteration in likelihood-ratio form, with the per-frame
 * fall-back rule described there) — which makes GPU-vs-oracle comparisons bit-exact on every frame.  Schedule,
 * recursion order and summation order are the reference's in both modes.
 */
#ifndef LDPC_ORACLE_H
#define LDPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MATH_LIBM = 0, ORC_MATH_DET = 1 };
enum { ORC_AWGN = 1, ORC_BSC = 2, ORC_BEC = 3 }; /* ldpcsim.h:15-20 */
#define ORC_ERASURE ((uint8_t)'E')                /* functions.h:105 */

typedef struct orc_code orc_code;
typedef struct orc_chan orc_chan;

/* ---- code file loader: ldpc.cpp:40-101, sparse.h:92-153 ---- */
orc_code *orc_code_load(const char *pc_file, const char *gen_file); /* NULL on open error */
void orc_code_free(orc_code *c);
int orc_code_nc(const orc_code *c);
int orc_code_mc(const orc_code *c);
int orc_code_kc(const orc_code *c);
int orc_code_nnz(const orc_code *c);
int orc_code_nct(const orc_code *c);
int orc_code_mct(const orc_code *c);
int orc_code_kct(const orc_code *c);
int orc_code_max_degree(const orc_code *c);
int orc_code_num_puncture(const orc_code *c);
int orc_code_num_shorten(const orc_code *c);
int orc_code_n_bitpos(const orc_code *c); /* == nct unless a column is both punctured and shortened */
int orc_code_has_G(const orc_code *c);
int orc_code_g_rows(const orc_code *c);
int orc_code_g_cols(const orc_code *c);
int orc_code_g_nnz(const orc_code *c);
/* copies: edge_row/edge_col [nnz] in file order; bit_pos [nct]; puncture/shorten */
void orc_code_edges(const orc_code *c, int *edge_row, int *edge_col);
void orc_code_bit_pos(const orc_code *c, int *bit_pos);
void orc_code_puncture(const orc_code *c, int *p);
void orc_code_shorten(const orc_code *c, int *s);
/* H * v (syndrome, sparse.h:201-211) and u * G accumulated into cw (sparse.h:163-172) */
void orc_syndrome(const orc_code *c, const uint8_t *word, uint8_t *synd);
void orc_encode_accumulate(const orc_code *c, const uint8_t *info, uint8_t *cw);
int orc_rank(const orc_code *c); /* GF(2) rank of H (dense elimination; value == sparse.h:233-300) */

/* ---- stand-alone decoders: decoder.cpp:11-78 and :91-192 ---- */
int orc_decode(const orc_code *c, int min_sum, int early_term, unsigned iterations, int math_mode,
               const double *llr_in, double *llr_out, uint8_t *hard);
/* mirrors of the opt-in NON-PARITY modes of the HIP library (1: flooding sum-product with binary32 messages, 2 / 3: layered
   schedule with binary32 / binary16 messages): n frames of given LLRs; tolerance comparisons only (see the .c file) */
void orc_decode_fast_batch(const orc_code *c, int mode, int early_term, unsigned iterations, uint64_t n, const double *llr_in,
                           uint32_t *iters, double *llr_out, uint8_t *hard);
/* step_of[row] = the step of the layered schedule the row is processed in; returns the number of steps */
int orc_layer_steps(const orc_code *c, int *step_of);
int orc_decode_bec(const orc_code *c, int early_term, unsigned iterations, int deg1_compat,
                   const uint8_t *llr_in, const uint8_t *codeword, uint8_t *llr_out, uint8_t *hard);

/* ORC_MATH_DET, sum-product with early termination: frames finished by the likelihood-ratio form / handed back
   to the LLR-domain form since the last reset (test introspection; single-threaded use only) */
void orc_ratio_stats(uint64_t *done, uint64_t *escaped, int reset);

/* ---- channel + decoder object: channel.cpp (one per reference OpenMP thread) ---- */
orc_chan *orc_chan_new(const orc_code *c, int chan_type, uint64_t seed, int math_mode,
                       int min_sum, int early_term, unsigned iterations, int bec_deg1_compat);
void orc_chan_free(orc_chan *ch);
void orc_chan_set_param(orc_chan *ch, double x); /* channel.cpp:37-42,123-127,193-197 */
void orc_chan_encode_and_map(orc_chan *ch);      /* channel.cpp:44-60 (only when G given) */
void orc_chan_simulate(orc_chan *ch);            /* channel.cpp:62-68,129-135,199-205 */
void orc_chan_calc_llrs(orc_chan *ch);           /* channel.cpp:70-93,137-162,207-229 */
int orc_chan_decode(orc_chan *ch);
const double *orc_chan_llr_in(const orc_chan *ch);
const double *orc_chan_llr_out(const orc_chan *ch);
const uint8_t *orc_chan_llr_in_bec(const orc_chan *ch);
const uint8_t *orc_chan_llr_out_bec(const orc_chan *ch);
const uint8_t *orc_chan_estimate(const orc_chan *ch);
const uint8_t *orc_chan_codeword(const orc_chan *ch);
uint64_t orc_chan_raw_draws(const orc_chan *ch); /* 64-bit draws taken from the noise stream since set_param */

/*
 * Run frames [0, skip+count) of the current channel point; frames < skip only advance the
 * RNG/codeword state (no decode).  Per decoded frame f (0-based after skip):
 *   iters[f], bit_errors[f] (ldpcsim.cpp:184-188, transmitted positions only) and, when the
 *   pointers are non-NULL, hard[f*nc..], llr_in[f*nc..], llr_out[f*nc..] (BEC: byte values
 *   widened to double), codeword[f*nc..].
 */
void orc_chan_run_frames(orc_chan *ch, uint64_t skip, uint64_t count, uint32_t *iters,
                         uint32_t *bit_errors, uint8_t *hard, double *llr_in, double *llr_out,
                         uint8_t *codeword);

/* ---- simulation loop: ldpcsim.cpp:97-263 ---- */
typedef struct
{
    double *fer, *ber, *avg_iter, *time;
    uint64_t *fec, *frames;
} orc_results;

/*
 * x-values MIN, MIN+STEP, ... < MAX (reversed for BSC/BEC); threads > 1 uses OpenMP with one
 * channel per thread seeded seed+tid, exactly as the reference.  results arrays must hold one
 * entry per x-value (written only when a frame error occurs, as in the reference);
 * totals[4*i..] = {frames, fec, bec, iters} is always written.  Returns the number of x-values.
 */
int orc_simulate(const orc_code *c, int chan_type, uint64_t seed, const double x_range[3],
                 int min_sum, int early_term, unsigned iterations, int math_mode,
                 int bec_deg1_compat, unsigned threads, uint64_t max_frames, uint64_t min_fec,
                 orc_results *results, uint64_t *totals, const volatile uint8_t *stop_flag);

/* raw RNG access for tests: first n outputs of mt19937_64(seed) */
void orc_mt64_stream(uint64_t seed, uint64_t n, uint64_t *out);
double orc_exp(int math_mode, double x);
double orc_log(int math_mode, double x);
/* element-by-element arithmetic for tests/test_gpu_math.py: what each function means (libm / long double), and
   detmath.h compiled for the host; fn as MathFn in libldpc_amd/csrc/kernels.hpp; 0 on success */
int orc_math_ref(int fn, uint64_t n, const double *a, const double *b, double *out);
int orc_math_det(int fn, uint64_t n, const double *a, const double *b, double *out);

#ifdef __cplusplus
}
#endif
#endif
