// engine.hpp — host runtime around the HIP kernels: device tables, workspaces, the mt19937_64
// stream engine and the batched channel+decode step.  One Engine per (code, GPU).
//
// This is the native replacement for the reference's per-thread `channel` objects
// (src/sim/channel.h:7-252) driven frame by frame from ldpc_sim::start (src/sim/ldpcsim.cpp:158-188):
// instead of five virtual calls per frame, one call decodes a batch of frames of the same stream.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "code.hpp"
#include "comm.hpp"
#include "kernels.hpp"
#include "mt64.hpp"
#include "plan.hpp"

namespace ldpc_amd
{

enum ChannelType : int
{
    kAwgn = 1, // same numbering as the reference's channel_type enum (ldpcsim.h:15-20)
    kBsc = 2,
    kBec = 3
};

struct DecParams
{
    bool early_term = true;
    uint32_t iterations = 50;
    bool min_sum = false;
};

// Output buffers of a batch; device or host pointers, nullptr = not wanted.
struct BatchOut
{
    uint32_t *iters = nullptr;      // [n]
    uint32_t *bit_errors = nullptr; // [n]
    uint8_t *hard = nullptr;        // [n][nc]
    double *llr_out = nullptr;      // [n][nc] (BEC: symbol values widened to double)
    double *llr_in = nullptr;       // [n][nc]
    uint8_t *codeword = nullptr;    // [n][nc]
};

class DeviceBuffer
{
  public:
    DeviceBuffer() = default;
    ~DeviceBuffer();
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    void *reserve(size_t bytes); // grow-only
    void *get() const { return ptr_; }
    size_t size() const { return size_; }

  private:
    void *ptr_ = nullptr;
    size_t size_ = 0;
};

// Bookkeeping of the chunk-start-state table of one stream (host only, no HIP calls: unit-testable without a GPU,
// ldpc_hip_selftest_chunk_table).  Row r of the device table holds the generator state at the start of chunk
// base()+r; rows [0, ready()) are valid; the table has kCap+1 rows (0..kCap).
struct ChunkTableOp
{
    enum Kind
    {
        kUploadWindow0, // row 0 := state of chunk 0 (from the seed)
        kRebase,        // row 0 := row a
        kJump           // rows [a, 2a) := rows [0, a) advanced by a chunks (polynomial index b: a == 1 << b)
    } kind;
    uint32_t a, b;
};

class ChunkTable
{
  public:
    static constexpr uint32_t kCap = 8192;
    void invalidate() { valid_ = false; }
    uint64_t base() const { return base_; }
    uint32_t ready() const { return ready_; }
    // append the operations that make the states of chunks [c_lo, c_hi) available as rows [c_lo-base, c_hi-base)
    void ensure(uint64_t c_lo, uint64_t c_hi, std::vector<ChunkTableOp> &ops);
    // row that may receive the state FOLLOWING chunk c_hi-1 when a generate launch over [c_lo, c_hi) produces it for
    // free, or -1 when the table has no row for it
    int64_t next_row(uint64_t c_hi) const { return c_hi - base_ <= kCap ? static_cast<int64_t>(c_hi - base_) : -1; }
    // the launch wrote next_row(c_hi)
    void note_next_written(uint64_t c_hi);

  private:
    bool valid_ = false;
    uint64_t base_ = 0;      // chunk id of row 0
    uint32_t ready_ = 0;     // rows [0, ready_) hold chunk start states
    uint32_t pow_ready_ = 0; // power-of-two prefix obtained by doubling
};

// page-locked host memory for small transfers (single-frame decode(): the copies are latency, not bandwidth)
class PinnedBuffer
{
  public:
    PinnedBuffer() = default;
    ~PinnedBuffer();
    PinnedBuffer(const PinnedBuffer &) = delete;
    PinnedBuffer &operator=(const PinnedBuffer &) = delete;
    void *reserve(size_t bytes); // grow-only
  private:
    void *ptr_ = nullptr;
    size_t size_ = 0;
};

// One mt19937_64(seed) stream: chunk start states by jump-ahead, raw words generated on demand.
class MtStream
{
  public:
    static constexpr uint64_t kChunkWords = 3360 * kMtWords; // 1048320 words = 8 MB per chunk
    static constexpr uint32_t kStateCap = ChunkTable::kCap;
    void reset(uint64_t seed);
    uint64_t seed() const { return seed_; }
    // chunks per generator workgroup (1 or 4, kernels.hpp launch_mt_generate)
    void set_pack(int chunks_per_workgroup) { pack_ = chunks_per_workgroup; }
    // Generate raw outputs [first, first+count) of the stream; returns a device pointer to word `first`.
    // raw words [first, first + count) on `stream`, into one of the stream object's two output buffers
    const uint64_t *generate(uint64_t first, uint64_t count, void *stream, int buffer = 0);

  private:
    void ensure_states(uint64_t c_lo, uint64_t c_hi, void *stream);
    const uint64_t *device_poly(unsigned m, void *stream);
    uint64_t seed_ = 0;
    int pack_ = 1;
    bool seeded_ = false;
    ChunkTable table_;
    unsigned polys_uploaded_ = 0;
    DeviceBuffer states_, raw_[2], poly_;
};

class Engine
{
  public:
    Engine(const std::string &pc_file, const std::string &gen_file, int device);
    ~Engine();

    const LdpcCode &code() const { return *code_; }
    const Plan &plan() const { return plan_; }
    const RegPlan &reg_plan() const { return reg_plan_; }
    const Reg2Plan &reg2_plan() const { return reg2_plan_; }
    int device() const { return device_; }
    bool bec_deg1_compat = false;
    // opt-in NON-PARITY fast mode: sum-product with binary32 messages (kernels_fast.hip); off by default
    bool fast_mode = false;

    // ---- decode given LLRs (C-ABI decode(), shared.cpp:47-65, batched) ----
    void decode_llr(const DecParams &p, uint64_t n, const double *llr_in, const BatchOut &out, void *stream);

    // ---- fused channel + decode on the reference's noise stream ----
    // set_channel_param semantics (channel.cpp:37-42): the stream restarts at frame 0.
    // `fresh` also resets the encoder (info-word stream position and the accumulated codeword), i.e. a
    // newly constructed channel object; the reference keeps both across channel points of one run.
    void stream_begin(int channel, uint64_t seed, double x, bool fresh = true);
    void stream_skip(uint64_t n_frames, void *stream);
    void stream_decode(const DecParams &p, uint64_t n_frames, const BatchOut &out, void *stream);
    // Un-consume the last `frames_back` frames of the most recent stream_decode batch from the ENCODER state
    // (info-word stream position and accumulated codeword), so that the next channel point continues where a
    // frame-by-frame run that stopped inside the batch would (the noise stream restarts per point anyway).
    void stream_rewind_encoder(uint64_t frames_back, void *stream);
    // ---- the same stream decoded by several ranks (one process per GPU), SURVEY §8e ----
    // One global step of about target_frames frames.  AWGN: the step is a fixed range of the RAW stream cut into
    // world equal pieces; rank r generates and scans only its piece (+ a margin of one frame's worth), the ranks
    // all-gather their accepted-pair counts (one u64 each), and a frame belongs to the rank whose piece holds its
    // first pair — so every frame keeps its place in mt19937_64(seed) and nobody scans another rank's noise.
    // BSC / BEC draw once per bit: frames split evenly, no exchange.  Every rank ends the step with the same
    // stream position.  `out` must hold shard_capacity(target_frames, world) frames.
    struct ShardStep
    {
        uint64_t step_first = 0, step_frames = 0; // the global step
        uint64_t first = 0, n = 0;                // this rank's frames [first, first + n)
    };
    static uint64_t shard_capacity(uint64_t target_frames, int world);
    ShardStep stream_decode_sharded(Comm &comm, const DecParams &p, uint64_t target_frames, const BatchOut &out, void *stream);
    // encoder state (info-word stream position + accumulated codeword) saved before a step / put back and advanced by
    // `frames` frames: how every rank lands on the state after the frame at which the simulation stopped
    void encoder_snapshot(void *stream);
    void encoder_restore_and_skip(uint64_t frames, void *stream);
    uint64_t stream_frame() const { return frame_pos_; }
    uint64_t stream_raw_draws() const;

    void synchronize(void *stream);
    uint64_t max_sub_batch() const; // frames one launch takes; larger requests are split

    // kernel timing with HIP events on the launch stream (bench.py's roofline figure)
    void set_profiling(bool on);
    // elapsed ms of the last batch: which = 0 decode kernel, 1 noise-stream kernels (generate + scan)
    float last_ms(int which);

  private:
    void bind_device();
    void ensure_rng_stream();
    // BSC / BEC: the batch's raw words generated on the engine's side stream into one of two buffers, ordered before the
    // caller's stream by an event — so that the generation for batch s+1 runs under the decode of batch s, as the AWGN
    // path's noise stream does; noise_raw_release() after the launch that reads them
    const uint64_t *noise_raw_async(uint64_t first, uint64_t count, void *stream, int &buffer);
    void noise_raw_release(int buffer, void *stream);
    void upload_plan();
    void run_decode(DecodeArgs &a, const DecParams &p, const BatchOut &out, uint64_t n, void *stream);
    void run_bec(const DecParams &p, const BatchOut &out, uint64_t n, const uint8_t *codeword, void *stream);
    void awgn_prepare(uint64_t n_frames, DecodeArgs &a, void *stream);
    // advance the encoder by n frames; returns the per-frame codewords [n][nc] (nullptr when no G is
    // loaded, or when want_codewords is false)
    const uint8_t *encode_frames(uint64_t n, bool want_codewords, void *stream);

    std::unique_ptr<LdpcCode> code_;
    Plan plan_;
    RegPlan reg_plan_;
    Reg2Plan reg2_plan_;
    DevPlan dev_{};
    DevRegPlan dev_reg_{};
    DevReg2Plan dev_reg2_{};
    int device_ = 0;
    bool device_checked_ = false;
    std::vector<void *> owned_;

    // stream state
    int chan_ = 0;
    double x_ = 0, sigma2_ = 0, sigma_ = 0, delta_ = 0;
    uint64_t frame_pos_ = 0;
    int stream_mode_ = 0; // 0 fresh, 1 stream_decode / stream_skip, 2 stream_decode_sharded (pair bookkeeping differs)
    uint64_t pair_next_ = 0; // accepted polar pairs located so far
    uint64_t raw_next_ = 0;  // raw draws consumed so far (AWGN: where the next trial starts)
    MtStream noise_, info_;
    uint64_t info_pos_ = 0;     // info-word draws consumed (kc per frame)
    bool cw_run_valid_ = false; // cw_run_ holds the accumulated codeword
    DeviceBuffer cw_run_, cw_next_, cw_frames_, cw_before_, enc_prefix_;
    uint64_t last_enc_n_ = 0;   // frames of the last encode_frames call that produced cw_frames_
    const uint32_t *g_col_ptr_ = nullptr, *g_col_row_ = nullptr;
    DeviceBuffer pairs_[2], carry_, scan_counts_, scan_offsets_, scan_result_;
    // the AWGN noise-stream kernels run on their own stream so that the pairs of batch s+1 are located while
    // the decode kernel of batch s drains; pairs_ is double-buffered, events order the two streams
    void *rng_stream_ = nullptr;
    void *ev_pairs_ready_[2] = {nullptr, nullptr}, *ev_pairs_free_[2] = {nullptr, nullptr};
    bool pairs_in_use_[2] = {false, false};
    int pp_ = 0;
    DeviceBuffer stage_in_, stage_iters_, stage_be_, stage_hard_, stage_llr_out_, stage_llr_in_, stage_cw_;
    DeviceBuffer ws_msg_, ws_llr_, ws_hb_, ws_scr_;
    PinnedBuffer pin_in_, pin_out_;
    void *pin_in_ev_ = nullptr; // hipEvent_t: the last host-to-device copy out of pin_in_
    DeviceBuffer enc_snap_;
    uint64_t enc_snap_pos_ = 0;
    bool enc_snap_valid_ = false;
    DeviceBuffer redo_; // [0] = count, [1..] = frames handed back by the ratio-form launch
    bool profiling_ = false;
    std::vector<void *> prof_pending_[2], prof_free_; // hipEvent_t: begin/end pairs per launch, spare events
    void prof_mark(int which, void *stream);
};

std::string hip_error_string(int err);

} // namespace ldpc_amd
