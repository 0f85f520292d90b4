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
#include "mtstates.hpp"
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

// The chunk start states of one mt19937_64(seed) stream on the device: executes the operations mtstates.hpp plans
// (uploads, copies, jump launches) and keeps the device copies of the jump polynomials.
class MtDevice
{
  public:
    MtDevice();
    ~MtDevice();
    MtDevice(const MtDevice &) = delete;
    MtDevice &operator=(const MtDevice &) = delete;
    void reset(uint64_t seed); // chunk states depend on the seed only: kept when it does not change
    uint64_t seed() const { return seed_; }
    uint32_t chunk_blocks() const { return chunk_blocks_; }
    // before the first use of the stream only (the engine picks the chunk size by the decoder's residency; LDPC_AMD_CHUNK_BLOCKS
    // overrides either choice)
    void set_default_chunk_blocks(uint32_t blocks)
    {
        if (!chunk_blocks_from_env_)
            chunk_blocks_ = blocks;
    }
    uint64_t chunk_words() const { return static_cast<uint64_t>(kMtWords) * chunk_blocks_; }
    uint64_t chunk_trials() const { return chunk_words() / 2; }
    // consecutive chunks [c_lo, c_hi): their start states in the ring; returns the ring row of c_lo
    uint32_t ensure_ring(uint64_t c_lo, uint64_t c_hi, void *stream);
    // look-ahead: the start states of the chunks up to c_hi computed NOW, on a stream of this object's own, so that the
    // jump-ahead chain of a later request runs beside the generator chain of an earlier one instead of in front of it
    // (ensure_ring waits for exactly the look-ahead launches whose rows it reads)
    void prefetch_ring(uint64_t c_hi);
    uint64_t *ring() const { return static_cast<uint64_t *>(ring_buf_.get()); }
    static constexpr uint32_t ring_rows() { return StateRing::kRows; }
    // sharded stream: rows 0 .. n-1 = chunks [first, first + n), consecutive calls `stride` chunks apart cost one launch
    uint64_t *ensure_strided(uint64_t first, uint32_t n, uint64_t stride, void *stream);
    static constexpr uint32_t strided_rows() { return StridedTable::kMaxRows; }
    // look-ahead: the launches that bring the sharded stream's tables of the next two steps into being, now, on the
    // object's own stream (three tables: the generator of this step reads one while the others are written)
    void prefetch_strided();
    uint64_t jump_tasks() const { return jump_tasks_; } // jump-ahead tasks launched so far (tools/shard_probe.py)
    void set_jump_pack(int tasks_per_workgroup) { jump_pack_ = tasks_per_workgroup; } // kernels.hpp launch_mt_jump
    void set_jump_groups(int groups) { jump_groups_ = groups; }

  private:
    void apply(const std::vector<StateOp> &ops, uint64_t *table, void *stream);
    const uint64_t *device_poly(uint64_t stride, void *stream);
    uint64_t seed_ = 0;
    bool seeded_ = false;
    uint32_t chunk_blocks_;
    bool chunk_blocks_from_env_ = false;
    StateRing ring_;
    StridedTable strided_;
    DeviceBuffer ring_buf_, strided_buf_;
    std::vector<std::pair<uint64_t, void *>> polys_; // (stride in chunks, device copy)
    uint64_t jump_tasks_ = 0;
    int jump_pack_ = 1;
    int jump_groups_ = 1;
    bool ring_polys_ready_ = false;
    // look-ahead launches in flight on jump_stream_: the ring rows of chunks [hi_before, hi_after) are valid after `event`
    struct Ahead
    {
        uint64_t hi_before, hi_after;
        void *event;
    };
    void drain_ahead(void *stream, uint64_t below); // `stream` waits for the look-ahead launches that wrote rows below `below`
    void *jump_stream_ = nullptr;
    void *ev_main_ = nullptr;   // recorded on the caller's stream after ring operations issued there
    bool main_dirty_ = false;   // ... which the look-ahead stream has not waited for yet
    std::vector<Ahead> ahead_;  // oldest first
    void *ev_strided_[StridedTable::kSlots] = {}; // after the look-ahead launch into a table of the sharded stream
    bool strided_pending_[StridedTable::kSlots] = {};
    std::vector<void *> ev_free_;
};

// Raw words of one mt19937_64(seed) stream (BSC / BEC draws, info words, ldpc_hip_mt64).
class MtStream
{
  public:
    void reset(uint64_t seed) { st.reset(seed), last_end_ = ~0ull; }
    uint64_t seed() const { return st.seed(); }
    // chunks per generator workgroup (1 or 4, kernels.hpp launch_mt_generate)
    void set_pack(int chunks_per_workgroup) { pack_ = chunks_per_workgroup; }
    int pack() const { return pack_; }
    // raw words [first, first + count) on `stream`, into one of the stream object's two output buffers; returns a device
    // pointer to word `first`
    const uint64_t *generate(uint64_t first, uint64_t count, void *stream, int buffer = 0);
    MtDevice st;

  private:
    int pack_ = 1;
    uint64_t last_end_ = ~0ull; // word after the previous request
    DeviceBuffer raw_[2];
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
    const FusedPlan &fused_plan() const { return fused_plan_; }
    int device() const { return device_; }
    bool bec_deg1_compat = false;
    // opt-in NON-PARITY modes, off (0) by default and never chosen by the library: 1 = flooding sum-product with binary32
    // messages (kernels_fast.hip); 2 / 3 = LAYERED sum-product with binary32 / binary16 messages (kernels_layered.hip)
    int fast_mode = 0;

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
    ShardStep stream_decode_sharded(Comm &comm, const DecParams &p, uint64_t target_frames, const BatchOut &out, void *stream,
                                    const std::string *failed_before = nullptr);
    // encoder state (info-word stream position + accumulated codeword) saved before a step / put back and advanced by
    // `frames` frames: how every rank lands on the state after the frame at which the simulation stopped
    void encoder_snapshot(void *stream);
    void encoder_restore_and_skip(uint64_t frames, void *stream);
    uint64_t stream_frame() const { return frame_pos_; }
    uint64_t stream_raw_draws(); // (AWGN: locates the last consumed trial with one generator pass over its chunk)
    // frames a sharded step of about target_frames can put on one rank (the output buffers' size)
    uint64_t shard_capacity(uint64_t target_frames, int world) const;
    uint64_t noise_jump_tasks() const { return noise_.st.jump_tasks(); }

    void synchronize(void *stream);
    uint64_t max_sub_batch() const; // frames one launch takes; larger requests are split

    // kernel timing with HIP events on the launch stream (bench.py's roofline figure)
    void set_profiling(bool on);
    // mean ms since the previous call: which = 0 decode launches, 1 noise-stream kernels (HIP events); 2 host time inside the
    // ranks' exchange, 3 host time waiting for the noise stream's result (wall clock)
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
    // normals of frames [frame_pos_, frame_pos_ + n): generate, count, place (write_normals false: count only, stream_skip)
    void awgn_prepare(uint64_t n_frames, DecodeArgs &a, void *stream, bool write_normals = true);
    // one generator pass over chunks [chunk, chunk + full) (+ a prefix of `last_blocks` blocks of the next one) on the side stream
    struct NoisePass
    {
        uint32_t n_slabs = 0;
        uint64_t *slabs = nullptr, *cum = nullptr;
        NormalsResult res{};
    };
    // Small requests in a row (one frame at a time: the reference's own call pattern) — the last single-chunk pass is kept:
    // it was generated a few frames further than asked, and while the following requests fall inside it they take their
    // normals from the slab that is already there instead of regenerating their chunk's prefix from its start
    struct SmallCache
    {
        bool valid = false;
        uint64_t chunk = 0;
        int buf = 0;
        NoisePass np;
    } small_cache_;
    NoisePass noise_pass(uint64_t chunk, uint32_t full, uint32_t last_blocks, uint32_t n_piece, uint64_t need, uint64_t target, int buf,
                         bool write_normals, bool strided, uint64_t stride);
    void fill_slab_args(DecodeArgs &a, const NoisePass &np, uint64_t pair_origin, int buf) const;
    // advance the encoder by n frames; returns the per-frame codewords [n][nc] (nullptr when no G is
    // loaded, or when want_codewords is false)
    const uint8_t *encode_frames(uint64_t n, bool want_codewords, void *stream);
    // the encoder's part of a sharded step of step_frames frames of which this rank decodes [before, before + n): the rank
    // draws the info words of ITS frames only, the ranks exchange the XOR of theirs (one all-gather of ceil(kc / 64) words),
    // and every rank ends with the codeword accumulated over the whole step.  Returns this rank's codewords [n][nc].
    const uint8_t *encode_frames_sharded(Comm &comm, uint64_t before, uint64_t n, uint64_t step_frames, void *stream);

    std::unique_ptr<LdpcCode> code_;
    Plan plan_;
    RegPlan reg_plan_;
    Reg2Plan reg2_plan_;
    bool shared6_ = false; // not LDS-resident and a check node of degree 6: three launches with early termination (dm_cn6_shared)
    FusedPlan fused_plan_; // fused form of the first ratio launch (kernels_fused.hip); ok = the code qualifies (fused_rule.h)
    DevFusedPlan dev_fused_{};
    LayerPlan layer_plan_;
    DevLayerPlan dev_layer_{};
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
    uint64_t raw_next_ = 0;  // BSC / BEC: raw draws consumed so far
    // AWGN, one rank reading the stream front to back: the pair with stream index cur_pair_ (= first normal of the next
    // frame >> 1) is the cur_k_-th accepted pair of chunk cur_chunk_
    uint64_t cur_pair_ = 0, cur_chunk_ = 0, cur_k_ = 0;
    // AWGN, sharded: the next step starts at chunk sh_chunk_, sh_pairs_ pairs were accepted before it
    uint64_t sh_chunk_ = 0, sh_pairs_ = 0;
    MtStream noise_, info_;
    uint64_t info_pos_ = 0;     // info-word draws consumed (kc per frame)
    bool cw_run_valid_ = false; // cw_run_ holds the accumulated codeword
    DeviceBuffer cw_run_, cw_next_, cw_frames_, cw_before_, enc_prefix_;
    uint64_t last_enc_n_ = 0;   // frames of the last encode_frames call that produced cw_frames_
    const uint32_t *g_col_ptr_ = nullptr, *g_col_row_ = nullptr;
    const uint64_t *g_mask_ = nullptr; // the columns of G as bit masks, [nc][words] (kernels.hpp, EncodeArgs::g_mask)
    DeviceBuffer slabs_[2], slab_cum_[2], nz_counts_, nz_result_, nz_locate_, nz_cum_skip_, nz_raw_, nz_lookback_;
    // the AWGN noise generator runs on its own stream so that the normals of batch s+1 are produced while
    // the decode kernel of batch s drains; the slabs are double-buffered, events order the two streams
    void *rng_stream_ = nullptr;
    void *ev_pairs_ready_[2] = {nullptr, nullptr}, *ev_pairs_free_[2] = {nullptr, nullptr};
    bool pairs_in_use_[2] = {false, false};
    int pp_ = 0;
    DeviceBuffer enc_base_; // sharded encoding: the other ranks' info-word sums (two rows of `words`)
    DeviceBuffer stage_in_, stage_iters_, stage_be_, stage_hard_, stage_llr_out_, stage_llr_in_, stage_cw_;
    DeviceBuffer ws_msg_, ws_llr_, ws_hb_, ws_scr_;
    PinnedBuffer pin_in_, pin_out_;
    void *pin_in_ev_ = nullptr; // hipEvent_t: the last host-to-device copy out of pin_in_
    DeviceBuffer enc_snap_;
    uint64_t enc_snap_pos_ = 0;
    bool enc_snap_valid_ = false;
    DeviceBuffer redo_, redo2_; // [0] = count, [1..] = frames handed back by the (first / second) ratio-form launch
    bool profiling_ = false;
    double host_ms_[2] = {0, 0}; // [0] exchange, [1] noise-stream wait
    uint64_t host_n_[2] = {0, 0};
    std::vector<void *> prof_pending_[2], prof_free_; // hipEvent_t: begin/end pairs per launch, spare events
    void prof_mark(int which, void *stream);
};

std::string hip_error_string(int err);

} // namespace ldpc_amd
