// engine.cpp — see engine.hpp.  Compiled by hipcc (host code using the HIP runtime API).
#include "engine.hpp"
#include "shard_place.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

namespace ldpc_amd
{

std::string hip_error_string(int err) { return hipGetErrorString(static_cast<hipError_t>(err)); }

namespace
{
// LDPC_AMD_TRACE=1: host wall-clock of the phases of a batch, on stderr
struct PhaseTrace
{
    bool on = std::getenv("LDPC_AMD_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
        if (!on)
            return;
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[ldpc_amd] %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

void check(hipError_t e, const char *what)
{
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
void check(int e, const char *what) { check(static_cast<hipError_t>(e), what); }

bool is_device_ptr(const void *p)
{
    if (!p)
        return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess)
    {
        (void)hipGetLastError(); // plain host memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// routes an output either straight to the caller's device pointer or through a staging buffer
struct OutStage
{
    struct Item
    {
        void *host, *dev;
        size_t bytes;
    };
    std::vector<Item> items;
    template <typename T>
    T *route(T *user, DeviceBuffer &stage, size_t bytes)
    {
        if (!user)
            return nullptr;
        if (is_device_ptr(user))
            return user;
        void *d = stage.reserve(bytes);
        items.push_back({user, d, bytes});
        return static_cast<T *>(d);
    }
    size_t pinned_bytes() const
    {
        size_t total = 0;
        for (auto &i : items)
            total += (i.bytes + 63) & ~size_t(63);
        return total;
    }
    // small results go through page-locked memory: a device-to-pageable copy is a blocking staged copy per call,
    // which is most of a single-frame decode()'s latency.  `word` (optional): a device word that rides along and is
    // returned — the results are delivered, the items kept, so that the caller can deliver them again after more launches.
    uint32_t flush(hipStream_t s, PinnedBuffer *pin = nullptr, const uint32_t *word = nullptr)
    {
        const size_t total = pinned_bytes();
        if (pin && total && total <= kPinnedLimit)
        {
            char *h = static_cast<char *>(pin->reserve(total + 64));
            size_t off = 0;
            for (auto &i : items)
            {
                check(hipMemcpyAsync(h + off, i.dev, i.bytes, hipMemcpyDeviceToHost, s), "copy out");
                off += (i.bytes + 63) & ~size_t(63);
            }
            if (word)
                check(hipMemcpyAsync(h + total, word, 4, hipMemcpyDeviceToHost, s), "copy out");
            check(hipStreamSynchronize(s), "sync");
            off = 0;
            for (auto &i : items)
            {
                std::memcpy(i.host, h + off, i.bytes);
                off += (i.bytes + 63) & ~size_t(63);
            }
            if (word)
            {
                uint32_t w;
                std::memcpy(&w, h + total, 4);
                return w;
            }
            items.clear();
            return 0;
        }
        if (word)
            throw std::runtime_error("OutStage: a ride-along word needs the page-locked path");
        for (auto &i : items)
            check(hipMemcpyAsync(i.host, i.dev, i.bytes, hipMemcpyDeviceToHost, s), "copy out");
        if (!items.empty())
            check(hipStreamSynchronize(s), "sync");
        items.clear();
        return 0;
    }
    static constexpr size_t kPinnedLimit = 1 << 20;
};

} // namespace

// ---------------------------------------------------------------------------------------------
PinnedBuffer::~PinnedBuffer()
{
    if (ptr_)
        (void)hipHostFree(ptr_);
}

void *PinnedBuffer::reserve(size_t bytes)
{
    if (bytes > size_)
    {
        if (ptr_)
            check(hipHostFree(ptr_), "hipHostFree");
        ptr_ = nullptr;
        const size_t want = std::max<size_t>(bytes, 64 * 1024);
        check(hipHostMalloc(&ptr_, want, hipHostMallocDefault), "hipHostMalloc");
        size_ = want;
    }
    return ptr_;
}

DeviceBuffer::~DeviceBuffer()
{
    if (ptr_)
        (void)hipFree(ptr_);
}

void *DeviceBuffer::reserve(size_t bytes)
{
    if (bytes > size_)
    {
        if (ptr_)
            check(hipFree(ptr_), "hipFree");
        ptr_ = nullptr;
        // an eighth of slack from the first allocation on: the per-batch buffers of the noise stream follow the number of
        // chunks a batch spans, which varies by one or two in eighty, and growing means hipFree — a wait for everything
        // the device has in flight, the decode kernel of the batch before included (it cost a 4 ms step 2 ms whenever a
        // batch set a new maximum: steps 1, 3, 52 and 64 of the headline run)
        size_t want = std::max(bytes + bytes / 8, size_ + size_ / 2);
        check(hipMalloc(&ptr_, want), "hipMalloc");
        size_ = want;
    }
    return ptr_;
}

// ---------------------------------------------------------------------------------------------
MtDevice::MtDevice()
{
    // 1120 blocks = 349 440 words = 2.8 MB of raw stream per chunk: the serial twist chain of a chunk has to hide under the
    // decode kernel, whose waves outrank it.  Round 3's headline kernel (3.6 ms) hid chunks of 2240 blocks; under round 4's
    // (2.6 ms) their chain outlasts it (step 2.95 ms), chunks of 840..1400 blocks all give 2.76..2.79 ms, and at 560 the
    // jump-ahead tasks (one per chunk, two steps ahead) take over.  A 65 536-frame batch spans 165 chunks.
    // LDPC_AMD_CHUNK_BLOCKS: experiments / tests of the chunk edges.
    chunk_blocks_ = 1120;
    if (const char *e = std::getenv("LDPC_AMD_CHUNK_BLOCKS"))
    {
        const long v = std::strtol(e, nullptr, 10);
        if (v >= 1 && v <= (1 << 20))
            chunk_blocks_ = static_cast<uint32_t>(v), chunk_blocks_from_env_ = true;
    }
}

MtDevice::~MtDevice()
{
    if (jump_stream_)
    {
        (void)hipStreamSynchronize(static_cast<hipStream_t>(jump_stream_));
        (void)hipStreamDestroy(static_cast<hipStream_t>(jump_stream_));
    }
    for (Ahead &a : ahead_)
        (void)hipEventDestroy(static_cast<hipEvent_t>(a.event));
    for (void *e : ev_free_)
        (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    if (ev_main_)
        (void)hipEventDestroy(static_cast<hipEvent_t>(ev_main_));
    for (void *e : ev_strided_)
        if (e)
            (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    for (auto &p : polys_)
        if (p.second)
            (void)hipFree(p.second);
}

void MtDevice::reset(uint64_t seed)
{
    if (seeded_ && seed == seed_)
        return; // chunk states depend on the seed only; keep them
    seed_ = seed;
    seeded_ = true;
    if (jump_stream_) // look-ahead launches of the old seed: done before the ring is written again
        check(hipStreamSynchronize(static_cast<hipStream_t>(jump_stream_)), "sync");
    for (Ahead &a : ahead_)
        ev_free_.push_back(a.event);
    ahead_.clear();
    for (bool &b : strided_pending_)
        b = false;
    ring_.invalidate();
    strided_.invalidate();
}

// device copy of t^(312 * chunk_blocks * stride) mod phi, uploaded once per stream object and stride
const uint64_t *MtDevice::device_poly(uint64_t stride, void *stream)
{
    for (auto &p : polys_)
        if (p.first == stride)
            return static_cast<const uint64_t *>(p.second);
    // bounded: beyond 64 device copies the older half goes, except the ring's powers of two (launches that read them may be
    // in flight: the device is drained first — once per 32 new step geometries of a simulation, if ever)
    if (polys_.size() >= 64)
    {
        check(hipDeviceSynchronize(), "sync before trimming the polynomial cache");
        std::vector<std::pair<uint64_t, void *>> keep;
        for (size_t i = 0; i < polys_.size(); ++i)
            if (i >= polys_.size() / 2 || (polys_[i].first & (polys_[i].first - 1)) == 0)
                keep.push_back(polys_[i]);
            else
                (void)hipFree(polys_[i].second);
        polys_.swap(keep);
    }
    const Gf2Poly &g = chunk_jump_poly(chunk_blocks_, stride);
    std::vector<uint64_t> padded(kJumpPolyWords, 0);
    std::copy(g.begin(), g.begin() + kMtWords, padded.begin());
    void *d = nullptr;
    check(hipMalloc(&d, sizeof(uint64_t) * kJumpPolyWords), "hipMalloc poly");
    polys_.emplace_back(stride, d);
    hipStream_t s = static_cast<hipStream_t>(stream);
    check(hipMemcpyAsync(d, padded.data(), sizeof(uint64_t) * kJumpPolyWords, hipMemcpyHostToDevice, s), "upload poly");
    check(hipStreamSynchronize(s), "sync"); // `padded` is a local buffer
    return static_cast<const uint64_t *>(d);
}

void MtDevice::apply(const std::vector<StateOp> &ops, uint64_t *table, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t row = sizeof(uint64_t) * kMtWords;
    for (const StateOp &op : ops)
        switch (op.kind)
        {
        case StateOp::kUpload0:
        {
            uint64_t w0[kMtWords];
            mt64_window0(seed_, w0);
            check(hipMemcpyAsync(table + static_cast<size_t>(op.dst) * kMtWords, w0, row, hipMemcpyHostToDevice, s), "upload window0");
            check(hipStreamSynchronize(s), "sync"); // w0 is a stack buffer
            break;
        }
        case StateOp::kCopy:
            check(hipMemcpyAsync(table + static_cast<size_t>(op.dst) * kMtWords, table + static_cast<size_t>(op.src) * kMtWords, row,
                                 hipMemcpyDeviceToDevice, s),
                  "state copy");
            break;
        case StateOp::kJump:
            check(launch_mt_jump(table, op.mod, op.src, op.dst, device_poly(op.stride, stream), op.n, jump_pack_, jump_groups_, s), "mt_jump");
            jump_tasks_ += op.n;
            break;
        }
}

uint32_t MtDevice::ensure_ring(uint64_t c_lo, uint64_t c_hi, void *stream)
{
    uint64_t *table = static_cast<uint64_t *>(ring_buf_.reserve(sizeof(uint64_t) * kMtWords * StateRing::kTotalRows));
    if (!ring_polys_ready_)
    {
        // every stride the ring will ever use (the powers of two up to its window), computed and uploaded NOW: a
        // polynomial squaring takes the host 10 ms, and the ring's window reaches its full width only after two dozen
        // batches — inside somebody's timed region
        for (uint32_t w = 1; w <= StateRing::kWindow; w *= 2)
            (void)device_poly(w, stream);
        ring_polys_ready_ = true;
    }
    std::vector<StateOp> ops;
    ring_.ensure(c_lo, c_hi, ops);
    // Look-ahead launches (prefetch_ring) in flight: the request waits for those that wrote a row it reads; operations
    // planned here (a seek, an extension the look-ahead did not cover) read and write rows anywhere: they wait for all.
    drain_ahead(stream, ops.empty() ? c_hi : ~0ull);
    apply(ops, table, stream);
    if (!ops.empty())
    {
        if (!ev_main_)
        {
            hipEvent_t e;
            check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
            ev_main_ = e;
        }
        check(hipEventRecord(static_cast<hipEvent_t>(ev_main_), static_cast<hipStream_t>(stream)), "event");
        main_dirty_ = true;
    }
    return static_cast<uint32_t>(c_lo % StateRing::kRows);
}

void MtDevice::drain_ahead(void *stream, uint64_t below)
{
    size_t n = 0;
    while (n < ahead_.size() && ahead_[n].hi_before < below)
        ++n;
    if (n == 0)
        return;
    // (one stream, in order: the newest of them implies the others)
    check(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ahead_[n - 1].event), 0), "wait look-ahead");
    for (size_t i = 0; i < n; ++i)
        ev_free_.push_back(ahead_[i].event); // (re-recorded only after this wait has been enqueued: safe to reuse)
    ahead_.erase(ahead_.begin(), ahead_.begin() + static_cast<std::ptrdiff_t>(n));
}

void MtDevice::prefetch_ring(uint64_t c_hi)
{
    if (!ring_.valid() || !ring_polys_ready_ || std::getenv("LDPC_AMD_NO_LOOKAHEAD"))
        return;
    const uint64_t hi_before = ring_.hi();
    std::vector<StateOp> ops;
    ring_.extend_to(c_hi, ops);
    if (ops.empty())
        return;
    if (!jump_stream_)
    {
        hipStream_t js;
        check(hipStreamCreateWithFlags(&js, hipStreamNonBlocking), "hipStreamCreate");
        jump_stream_ = js;
    }
    hipStream_t js = static_cast<hipStream_t>(jump_stream_);
    if (main_dirty_) // rows written on the caller's stream (the seek, the first extensions) are sources here
    {
        check(hipStreamWaitEvent(js, static_cast<hipEvent_t>(ev_main_), 0), "wait ring");
        main_dirty_ = false;
    }
    apply(ops, ring(), js);
    void *ev;
    if (!ev_free_.empty())
        ev = ev_free_.back(), ev_free_.pop_back();
    else
    {
        hipEvent_t e;
        check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
        ev = e;
    }
    check(hipEventRecord(static_cast<hipEvent_t>(ev), js), "event");
    ahead_.push_back({hi_before, ring_.hi(), ev});
}

uint64_t *MtDevice::ensure_strided(uint64_t first, uint32_t n, uint64_t stride, void *stream)
{
    uint64_t *table = static_cast<uint64_t *>(strided_buf_.reserve(sizeof(uint64_t) * kMtWords * StridedTable::kTotalRows));
    std::vector<StateOp> ops;
    const uint32_t base = strided_.position(first, n, stride, ops);
    // a look-ahead launch wrote the table that is read now; operations planned here may touch any table: they wait for all
    for (uint32_t k = 0; k < StridedTable::kSlots; ++k)
        if (strided_pending_[k] && (k == base / StridedTable::kMaxRows || !ops.empty()))
        {
            check(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_strided_[k]), 0), "wait look-ahead");
            strided_pending_[k] = false;
        }
    apply(ops, table, stream);
    if (!ops.empty())
    {
        if (!ev_main_)
        {
            hipEvent_t e;
            check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
            ev_main_ = e;
        }
        check(hipEventRecord(static_cast<hipEvent_t>(ev_main_), static_cast<hipStream_t>(stream)), "event");
        main_dirty_ = true;
    }
    return table + static_cast<size_t>(base) * kMtWords;
}

void MtDevice::prefetch_strided()
{
    if (std::getenv("LDPC_AMD_NO_LOOKAHEAD") || !strided_buf_.get())
        return;
    std::vector<StateOp> ops;
    strided_.look_ahead(ops);
    if (ops.empty())
        return;
    if (!jump_stream_)
    {
        hipStream_t js;
        check(hipStreamCreateWithFlags(&js, hipStreamNonBlocking), "hipStreamCreate");
        jump_stream_ = js;
    }
    hipStream_t js = static_cast<hipStream_t>(jump_stream_);
    if (main_dirty_)
    {
        check(hipStreamWaitEvent(js, static_cast<hipEvent_t>(ev_main_), 0), "wait table");
        main_dirty_ = false;
    }
    for (const StateOp &op : ops) // one launch per table, its event for the step that reads that table
    {
        apply({op}, static_cast<uint64_t *>(strided_buf_.get()), js);
        const uint32_t k = op.dst / StridedTable::kMaxRows;
        if (!ev_strided_[k])
        {
            hipEvent_t e;
            check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
            ev_strided_[k] = e;
        }
        check(hipEventRecord(static_cast<hipEvent_t>(ev_strided_[k]), js), "event");
        strided_pending_[k] = true;
    }
}

const uint64_t *MtStream::generate(uint64_t first, uint64_t count, void *stream, int buffer)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (count == 0)
        count = 1;
    const uint64_t cw = st.chunk_words();
    const uint64_t c_lo = first / cw;
    const uint64_t c_hi = (first + count + cw - 1) / cw;
    if (c_hi - c_lo > StateRing::kWindow)
        throw std::runtime_error("mt19937_64 stream request exceeds the chunk-state window");
    const uint32_t row = st.ensure_ring(c_lo, c_hi, stream);
    const uint32_t n = static_cast<uint32_t>(c_hi - c_lo);
    uint64_t *raw = static_cast<uint64_t *>(raw_[buffer & 1].reserve(sizeof(uint64_t) * cw * n));
    uint32_t words = static_cast<uint32_t>(cw);
    if (n == 1)
    {
        // a short request inside one chunk (single frames, small batches): only the prefix of the chunk that is asked
        // for is generated (whole twist rounds of 312 words)
        const uint64_t need = (first + count - c_lo * cw + kMtWords - 1) / kMtWords * kMtWords;
        words = static_cast<uint32_t>(std::min<uint64_t>(need, cw));
    }
    check(launch_mt_generate(st.ring(), MtDevice::ring_rows(), row, raw, n, words, words, n == 1 ? 1 : pack_, s), "mt_generate");
    // a reader going through the stream front to back (one rank: every request starts where the one before ended): the
    // chunk start states of the next request but one, now (MtDevice::prefetch_ring).  A rank of a sharded stream reads its
    // share of every step, a fixed distance apart: nothing to look ahead for, the ring moves by that distance in one launch.
    if (first == last_end_ && n > 1)
        st.prefetch_ring(c_hi + 2 * (static_cast<uint64_t>(n) + 1));
    last_end_ = first + count;
    return raw + (first - c_lo * cw);
}

// ---------------------------------------------------------------------------------------------
Engine::Engine(const std::string &pc_file, const std::string &gen_file, int device) : device_(device)
{
    code_ = std::make_unique<LdpcCode>(pc_file, gen_file);
    if (code_->min_cn_degree() < 2)
        throw std::runtime_error("check nodes of degree < 2 are not supported (undefined in the reference decoder)");
    plan_ = build_plan(*code_);
    if (plan_.lds_ok && !std::getenv("LDPC_AMD_NO_FUSED")) // (the variable: experiments only — results change by ulps)
        fused_plan_ = build_fused_plan(*code_, plan_);
    if (!plan_.lds_ok)
        for (int r = 0; r < code_->H.rows && !shared6_; ++r)
            shared6_ = code_->H.rptr[r + 1] - code_->H.rptr[r] == 6;
    if (!plan_.lds_ok) // register-resident decoder: the smallest register tile the code fits
    {
        // One frame per CU (1024 threads, 128 VGPRs, 160 KB mailbox) first: measured 1.37x faster on the n=8192
        // code than two frames per CU (512 threads, 256 VGPRs, 80 KB mailboxes, one more exchange round), which
        // remains available for tiles that do not fit 128 registers (LDPC_AMD_REG_NT512=1 forces it).
        struct Tile { int nt, kc, maxd; };
        const bool two_per_cu = std::getenv("LDPC_AMD_REG_NT512") != nullptr;
        for (Tile t : {Tile{1024, 4, 6}, Tile{1024, 8, 4}, Tile{1024, 2, 8}, Tile{512, 8, 6}, Tile{512, 16, 4}, Tile{512, 4, 8}})
        {
            if (two_per_cu && t.nt != 512)
                continue;
            reg_plan_ = build_reg_plan(*code_, plan_, t.nt, t.kc, t.maxd, t.nt == 512 ? 80 * 1024 : 160 * 1024);
            if (reg_plan_.ok)
                break;
        }
        // second form (totals come back instead of messages): preferred when the code fits its one instantiation
        if (reg_plan_.ok && !std::getenv("LDPC_AMD_NO_REG2"))
            reg2_plan_ = build_reg2_plan(*code_, plan_, 1024, 4, 6, 4, 4);
        // a register-resident decode workgroup owns its CU: keep the noise generator of the next batch, which runs
        // beside it, on a quarter of the CUs (config 4: 4.89 -> 4.62 ms per step)
        // ... and in chunks of 2240 blocks: half as many jump-ahead tasks per batch as the LDS-resident decoders' 1120 — here
        // every task holds a CU that a frame could have, and the generator's longer chain still ends before the 4.6 ms launch
        if (reg_plan_.ok)
            noise_.set_pack(4), noise_.st.set_jump_pack(3), noise_.st.set_default_chunk_blocks(2240);
    }
}

Engine::~Engine()
{
    for (void *p : owned_)
        (void)hipFree(p);
    for (auto &q : prof_pending_)
        for (void *e : q)
            prof_free_.push_back(e);
    for (void *e : prof_free_)
        if (e)
            (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    for (int i = 0; i < 2; ++i)
    {
        if (ev_pairs_ready_[i])
            (void)hipEventDestroy(static_cast<hipEvent_t>(ev_pairs_ready_[i]));
        if (ev_pairs_free_[i])
            (void)hipEventDestroy(static_cast<hipEvent_t>(ev_pairs_free_[i]));
    }
    if (rng_stream_)
        (void)hipStreamDestroy(static_cast<hipStream_t>(rng_stream_));
    if (pin_in_ev_)
        (void)hipEventDestroy(static_cast<hipEvent_t>(pin_in_ev_));
}

void Engine::set_profiling(bool on) { profiling_ = on; }

// Every public entry point binds the calling thread to this engine's GPU: the current device is per-thread state, and
// a context used from a new thread (or after another engine switched devices) would otherwise launch on device 0 with
// this device's pointers.
void Engine::bind_device()
{
    if (!device_checked_)
    {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= device_)
        {
            (void)hipGetLastError();
            throw std::runtime_error("no usable HIP device (MI355X required): " + std::string(hipGetErrorString(e)));
        }
        device_checked_ = true;
    }
    check(hipSetDevice(device_), "hipSetDevice");
}

// Event pairs are queued per launch and only read back in last_ms(), so that profiling does not serialise
// the noise stream of batch s+1 behind the decode of batch s.
void Engine::prof_mark(int which, void *stream)
{
    if (!profiling_)
        return;
    auto &q = prof_pending_[which];
    if (q.size() >= 8192 && q.size() % 2 == 0) // nobody is reading: keep the queue bounded
    {
        prof_free_.insert(prof_free_.end(), q.begin(), q.end());
        q.clear();
    }
    void *e;
    if (prof_free_.empty())
    {
        hipEvent_t ev;
        check(hipEventCreate(&ev), "hipEventCreate");
        e = ev;
    }
    else
    {
        e = prof_free_.back();
        prof_free_.pop_back();
    }
    check(hipEventRecord(static_cast<hipEvent_t>(e), static_cast<hipStream_t>(stream)), "event");
    q.push_back(e);
}

// mean duration (ms) of the launches of kind `which` (0 decode kernel, 1 noise stream) since the previous call
float Engine::last_ms(int which)
{
    if (which == 2 || which == 3)
    {
        const int k = which - 2;
        const float v = host_n_[k] ? static_cast<float>(host_ms_[k] / static_cast<double>(host_n_[k])) : 0.f;
        host_ms_[k] = 0, host_n_[k] = 0;
        return v;
    }
    bind_device();
    auto &q = prof_pending_[which ? 1 : 0];
    double sum = 0;
    size_t spans = 0;
    for (size_t i = 0; i + 1 < q.size(); i += 2)
    {
        hipEvent_t a = static_cast<hipEvent_t>(q[i]), b = static_cast<hipEvent_t>(q[i + 1]);
        float ms = 0.f;
        if (hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess)
            sum += ms, ++spans;
        else
            (void)hipGetLastError();
    }
    for (void *e : q)
        prof_free_.push_back(e);
    q.clear();
    return spans ? static_cast<float>(sum / spans) : 0.f;
}

void Engine::upload_plan()
{
    bind_device();
    if (dev_.cn_blocks)
        return;
    auto up = [&](const void *src, size_t bytes) -> void * {
        void *d = nullptr;
        check(hipMalloc(&d, std::max<size_t>(bytes, 16)), "hipMalloc plan");
        owned_.push_back(d);
        if (bytes)
            check(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice), "upload plan");
        return d;
    };
    const Plan &p = plan_;
    dev_.nc = p.nc, dev_.mc = p.mc, dev_.nnz = p.nnz, dev_.nct = p.nct;
    dev_.n_bitpos = p.n_bitpos;
    dev_.n_cn_blocks = static_cast<int>(p.cn_blocks.size());
    dev_.n_vn_blocks = static_cast<int>(p.vn_blocks.size());
    dev_.cn_work_stride = p.cn_work_stride, dev_.vn_work_stride = p.vn_work_stride;
#define UP(field, vec) dev_.field = static_cast<decltype(dev_.field)>(up(vec.data(), vec.size() * sizeof(vec[0])))
    UP(cn_blocks, p.cn_blocks);
    UP(vn_blocks, p.vn_blocks);
    UP(vn_slot, p.vn_slot);
    UP(cn_work, p.cn_work);
    UP(cn_work_desc, p.cn_work_desc);
    UP(vn_work_desc, p.vn_work_desc);
    if (!p.vn_packed.empty())
        UP(vn_packed, p.vn_packed);
    else
        dev_.vn_packed = nullptr;
    dev_.cn_desc_stride = p.cn_desc_stride;
    UP(vn_work, p.vn_work);
    UP(col_rank, p.col_rank);
    UP(rank_col, p.rank_col);
    UP(tx_rank, p.tx_rank);
    UP(rank_kind, p.rank_kind);
    UP(rank_slot0, p.rank_slot0);
    UP(bit_pos, code_->bit_pos);
#undef UP
    if (reg_plan_.ok)
    {
        const RegPlan &r = reg_plan_;
        dev_reg_.nt = r.nt, dev_reg_.kc = r.kc, dev_reg_.maxd = r.maxd, dev_reg_.rounds = r.rounds;
        dev_reg_.mb_doubles = r.mb_doubles;
        dev_reg_.cn_edge = static_cast<const uint32_t *>(up(r.cn_edge.data(), r.cn_edge.size() * 4));
        dev_reg_.cn_deg = static_cast<const uint8_t *>(up(r.cn_deg.data(), r.cn_deg.size()));
        dev_reg_.cn_cnt = static_cast<const uint8_t *>(up(r.cn_cnt.data(), r.cn_cnt.size()));
        dev_reg_.vn_blocks = static_cast<const RegVnBlock *>(up(r.vn_blocks.data(), r.vn_blocks.size() * sizeof(RegVnBlock)));
        dev_reg_.round_first = static_cast<const uint32_t *>(up(r.round_first.data(), r.round_first.size() * 4));
    }
    if (reg2_plan_.ok)
    {
        const Reg2Plan &r = reg2_plan_;
        dev_reg2_.nt = r.nt, dev_reg2_.kc = r.kc, dev_reg2_.maxd = r.maxd, dev_reg2_.nv0 = r.nv0, dev_reg2_.nv1 = r.nv1;
        dev_reg2_.neutral = r.neutral, dev_reg2_.lds_entries = r.lds_entries;
        dev_reg2_.uniform_cn = r.uniform_cn ? 1 : 0;
        dev_reg2_.uniform_vn = r.uniform_vn ? 1 : 0;
        std::memcpy(dev_reg2_.vn_affine, r.vn_affine, sizeof r.vn_affine);
        dev_reg2_.edge_w = static_cast<const uint32_t *>(up(r.edge_w.data(), r.edge_w.size() * 4));
        dev_reg2_.cn_deg = static_cast<const uint8_t *>(up(r.cn_deg.data(), r.cn_deg.size()));
        dev_reg2_.vn_blocks = static_cast<const Reg2VnBlock *>(up(r.vn_blocks.data(), r.vn_blocks.size() * sizeof(Reg2VnBlock)));
        dev_reg2_.vn_rank = static_cast<const uint32_t *>(up(r.vn_rank.data(), r.vn_rank.size() * 4));
    }
    if (fused_plan_.ok)
    {
        const FusedPlan &f = fused_plan_;
        dev_fused_.n_slots = f.n_slots, dev_fused_.vnb = f.vnb, dev_fused_.cnl = f.cnl, dev_fused_.calls_stride = f.calls_stride;
        dev_fused_.has_shortened = f.has_shortened ? 1 : 0;
        dev_fused_.need_lambda = f.need_lambda ? 1 : 0;
        dev_fused_.wide_exclusive = f.wide_exclusive ? 1 : 0;
        std::memcpy(dev_fused_.vn_prog, f.vn_prog, sizeof f.vn_prog);
        // the message slots; the prologue stages one 16-byte entry per transmitted bit or column (+ 2) in the same space
        dev_fused_.lds_bytes = static_cast<uint32_t>(std::max<size_t>(8 * static_cast<size_t>(f.n_slots), 16 * (static_cast<size_t>(std::max(p.nc, p.nct)) + 2)) + 15) & ~15u;
        dev_fused_.leaf_calls = static_cast<const FusedCall *>(up(f.leaf_calls.data(), f.leaf_calls.size() * sizeof(FusedCall)));
        dev_fused_.calls = static_cast<const FusedCall *>(up(f.calls.data(), f.calls.size() * sizeof(FusedCall)));
        dev_fused_.vn_desc = static_cast<const uint32_t *>(up(f.vn_desc.data(), f.vn_desc.size() * 4));
        dev_fused_.vn_slot = static_cast<const uint32_t *>(up(f.vn_slot.data(), f.vn_slot.size() * 4));
        dev_fused_.lane_tab = static_cast<const uint32_t *>(up(f.lane_tab.data(), f.lane_tab.size() * 4));
        dev_fused_.ho_map = static_cast<const uint32_t *>(up(f.ho_map.data(), f.ho_map.size() * 4));
    }
    if (code_->has_G())
    {
        std::vector<uint32_t> cp(code_->G.cptr.begin(), code_->G.cptr.end()), cr(code_->G.crow.begin(), code_->G.crow.end());
        g_col_ptr_ = static_cast<const uint32_t *>(up(cp.data(), cp.size() * 4));
        g_col_row_ = static_cast<const uint32_t *>(up(cr.data(), cr.size() * 4));
        const int kc = code_->kc(), words = (kc + 63) / 64;
        if (kc > 0 && words <= 4 && code_->G.rows <= kc)
        {
            std::vector<uint64_t> mask(static_cast<size_t>(p.nc) * words, 0);
            for (int j = 0; j < code_->G.cols && j < p.nc; ++j)
                for (int q = code_->G.cptr[j]; q < code_->G.cptr[j + 1]; ++q)
                {
                    const int r = code_->G.crow[q];
                    mask[static_cast<size_t>(j) * words + r / 64] ^= 1ull << (r % 64); // (a repeated entry cancels, as in the walk)
                }
            g_mask_ = static_cast<const uint64_t *>(up(mask.data(), mask.size() * 8));
        }
    }
    dev_.lds_bytes = static_cast<uint32_t>(p.lds_bytes);
}

void Engine::synchronize(void *stream)
{
    bind_device();
    check(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "sync");
}

// frames per launch: bounded so that the noise-stream buffers and the memory-resident workspace stay modest
uint64_t Engine::max_sub_batch() const
{
    // the normals of a batch: at most kMaxSlabs chunk slabs (4 GB) per buffer
    const uint64_t kMaxSlabs = std::max<uint64_t>(4, 476ull * 2240 / noise_.st.chunk_blocks());
    const uint64_t pairs_per_frame = std::max<uint64_t>(1, (static_cast<uint64_t>(plan_.nct) + 1) / 2 + 1);
    const uint64_t by_noise = std::max<uint64_t>(1, (kMaxSlabs - 2) * noise_.st.chunk_trials() * 3 / 4 / pairs_per_frame);
    if (plan_.lds_ok || reg_plan_.ok)
        return std::min<uint64_t>(1u << 17, by_noise);
    const uint64_t per_frame = 8ull * plan_.nnz + 8ull * plan_.nc + plan_.nnz;
    return std::max<uint64_t>(1, std::min<uint64_t>({1u << 17, (8ull << 30) / per_frame, by_noise}));
}

void Engine::run_decode(DecodeArgs &a, const DecParams &p, const BatchOut &out, uint64_t n, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nc = plan_.nc, nnz = plan_.nnz;
    OutStage st;
    a.plan = dev_;
    a.iterations = p.iterations;
    a.early_term = p.early_term;
    a.n_frames = n;
    a.iters = st.route(out.iters, stage_iters_, 4 * n);
    a.bit_errors = st.route(out.bit_errors, stage_be_, 4 * n);
    a.hard = st.route(out.hard, stage_hard_, n * nc);
    a.llr_out = st.route(out.llr_out, stage_llr_out_, 8 * n * nc);
    a.llr_in_dump = st.route(out.llr_in, stage_llr_in_, 8 * n * nc);
    prof_mark(0, s);
    if (fast_mode && !p.min_sum)
    {
        // the caller asked for a non-parity mode; never chosen by itself
        if (fast_mode == 1)
        {
            if (!fast_mode_supported(dev_, plan_.max_cn_degree) || plan_.has_isolated_vn)
                throw std::runtime_error("fast mode: this code is outside what the binary32 kernel takes (check nodes up to degree 8, "
                                         "nc <= 8192, LDS-resident, no isolated variable node)");
            a.ws_hb = static_cast<uint8_t *>(ws_hb_.reserve(n * nc));
            check(launch_decode_fast(a, plan_.max_cn_degree, s), "decode (fast mode, binary32)");
        }
        else
        {
            if (!layer_plan_.ok && layer_plan_.steps.empty())
            {
                layer_plan_ = build_layer_plan(*code_, plan_);
                if (layer_plan_.ok)
                {
                    std::vector<uint32_t> st;
                    for (const LayerStep &l : layer_plan_.steps)
                        st.push_back(l.off), st.push_back(static_cast<uint32_t>(l.count) | static_cast<uint32_t>(l.degree) << 16);
                    // the steps' neighbour tables as the kernel fetches them: four words per lane, two VN ranks per word
                    std::vector<uint32_t> vn4(layer_plan_.steps.size() * 4 * kWaveSize, 0);
                    for (size_t si = 0; si < layer_plan_.steps.size(); ++si)
                        for (int j = 0; j < layer_plan_.steps[si].degree; ++j)
                            for (int l = 0; l < kWaveSize; ++l)
                                vn4[(si * 4 + j / 2) * kWaveSize + l] |=
                                    static_cast<uint32_t>(layer_plan_.vn[layer_plan_.steps[si].off + static_cast<size_t>(j) * kWaveSize + l]) << (16 * (j & 1));
                    void *d_steps = nullptr, *d_vn = nullptr;
                    check(hipMalloc(&d_steps, st.size() * 4), "hipMalloc layer plan");
                    owned_.push_back(d_steps);
                    check(hipMalloc(&d_vn, vn4.size() * 4), "hipMalloc layer plan");
                    owned_.push_back(d_vn);
                    check(hipMemcpy(d_steps, st.data(), st.size() * 4, hipMemcpyHostToDevice), "upload layer plan");
                    check(hipMemcpy(d_vn, vn4.data(), vn4.size() * 4, hipMemcpyHostToDevice), "upload layer plan");
                    dev_layer_.steps = static_cast<const uint32_t *>(d_steps);
                    dev_layer_.vn4 = static_cast<const uint32_t *>(d_vn);
                    dev_layer_.n_steps = static_cast<uint32_t>(layer_plan_.steps.size());
                    dev_layer_.slots = layer_plan_.slots;
                    const size_t tot_bytes = 4 * ((nc + 3) & ~size_t(3));
                    dev_layer_.region_bytes = static_cast<uint32_t>((std::max(8 * nc, tot_bytes + 4 * size_t(layer_plan_.slots)) + 15) & ~size_t(15));
                    dev_layer_.region_bytes_half = static_cast<uint32_t>((std::max(8 * nc, tot_bytes + 2 * size_t(layer_plan_.slots)) + 15) & ~size_t(15));
                }
            }
            const bool half = fast_mode == 3;
            if (!layer_plan_.ok || plan_.has_isolated_vn || (half ? dev_layer_.region_bytes_half : dev_layer_.region_bytes) > 160 * 1024)
                throw std::runtime_error("layered mode: this code is outside what the layered kernel takes (check nodes of degree 2..8, at "
                                         "most 65535 columns, totals and messages of one frame within 160 KB of LDS, no isolated variable node)");
            check(launch_decode_layered(a, dev_layer_, half, s), half ? "decode (layered, binary16 messages)" : "decode (layered, binary32 messages)");
        }
        prof_mark(0, s);
        if (a.mode == kModeAwgn && a.pairs_buffer >= 0 && ev_pairs_free_[a.pairs_buffer])
        {
            check(hipEventRecord(static_cast<hipEvent_t>(ev_pairs_free_[a.pairs_buffer]), s), "event");
            pairs_in_use_[a.pairs_buffer] = true;
        }
        if (out.codeword)
        {
            if (a.codeword)
                check(hipMemcpyAsync(out.codeword, a.codeword, n * nc,
                                     is_device_ptr(out.codeword) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s), "codeword out");
            else if (is_device_ptr(out.codeword))
                check(hipMemsetAsync(out.codeword, 0, n * nc, s), "codeword out");
            else
                std::memset(out.codeword, 0, n * nc);
        }
        st.flush(s, &pin_out_);
        return;
    }
    bool fused_handover_used = false;
    bool finished_in_one = false; // the launch was the totals-form register kernel's chain of all three forms (kernels_reg2_impl.hpp)
    const auto launch = [&] {
        // the first launch of sum-product with early termination, for codes the fused form takes (fused_rule.h)
        if (fused_plan_.ok && !p.min_sum && p.early_term && a.redo_list && !a.redo_count_in && !a.ratio_separate)
            check(launch_decode_fused(a, dev_fused_, s), "decode (fused form)");
        else if (fused_plan_.ok && !p.min_sum && !p.early_term && a.redo_list && a.redo_iter && !a.redo_count_in && !std::getenv("LDPC_AMD_NO_FUSED_HO"))
        {
            // without early termination: the fused form with separately divided outputs until a frame's totals near the edge
            // of the box, then the LLR-domain launch below continues it (the messages are handed over as LLRs)
            check(launch_decode_fused_handover(a, dev_fused_, s), "decode (fused form, hand-over)");
            fused_handover_used = true;
        }
        else if (fused_plan_.ok && p.min_sum && !p.early_term && p.iterations > 0 && !a.redo_list && !a.redo_count_in &&
                 !std::getenv("LDPC_AMD_NO_FUSED_MS"))
            check(launch_decode_fused_minsum(a, dev_fused_, s), "decode (min-sum, fused plan)");
        else if (plan_.lds_ok)
        {
            // Input LLRs: in registers when that frees the LDS for one more resident frame per CU (n=1024 code:
            // 40 KB -> 31 KB, five frames instead of four) and the plan allows it; in LDS otherwise.  Device memory
            // (mode 1) also reaches five frames but pays for it in memory reads (measured: no net gain).
            const size_t cu_lds = 160 * 1024, with_llr = plan_.lds_bytes, without = plan_.lds_bytes - 8 * nc;
            int llr_mode = 0;
            if (cu_lds / without > cu_lds / with_llr && plan_.vn_work_stride <= 8 && !plan_.has_isolated_vn &&
                plan_.nc <= plan_.nnz && !plan_.vn_packed.empty())
                llr_mode = 2;
            if (const char *e = std::getenv("LDPC_AMD_LLR_MODE"))
                llr_mode = std::atoi(e);
            if (llr_mode == 1)
                a.ws_llr = static_cast<double *>(ws_llr_.reserve(8 * n * nc));
            if (const char *e = std::getenv("LDPC_AMD_LDS_PAD")) // occupancy experiments: extra dynamic LDS per frame
                a.plan.lds_bytes = dev_.lds_bytes + (static_cast<uint32_t>(std::strtoul(e, nullptr, 10)) & ~15u);
            check(launch_decode_lds(a, p.min_sum, plan_.max_cn_degree, llr_mode, s), "decode (LDS-resident)");
        }
        else if (reg_plan_.ok && !std::getenv("LDPC_AMD_NO_REG"))
        {
            a.ws_llr = static_cast<double *>(ws_llr_.reserve(8 * n * nc));
            a.ws_hb = static_cast<uint8_t *>(ws_hb_.reserve(n * nc));
            if (reg2_plan_.ok)
            {
                // channel terms of the variable nodes, one per (block slot, thread): kernels_reg2.hip
                a.ws_scr = static_cast<double *>(ws_scr_.reserve(8 * n * static_cast<uint64_t>(reg2_plan_.nv0 + reg2_plan_.nv1) * reg2_plan_.nt));
                check(launch_decode_reg2(a, dev_reg2_, p.min_sum, s), "decode (register-resident, totals form)");
                finished_in_one = a.redo_list != nullptr;
            }
            else
                check(launch_decode_reg(a, dev_reg_, p.min_sum, s), "decode (register-resident)");
        }
        else if (plan_.hbm_ok)
        {
            a.ws_msg = static_cast<double *>(ws_msg_.reserve(8 * n * nnz));
            a.ws_llr = static_cast<double *>(ws_llr_.reserve(8 * n * nc));
            a.ws_hb = static_cast<uint8_t *>(ws_hb_.reserve(n * nnz));
            a.ws_scr = plan_.max_cn_degree > kMaxCnDegree ? static_cast<double *>(ws_scr_.reserve(16 * n * nnz)) : nullptr;
            // resident frames per CU are bounded through a dummy LDS request so that the frames in flight
            // (256 CUs x frames/CU x state bytes) stay inside the 256 MiB Infinity Cache
            const uint64_t per_frame = 8ull * nnz + 8ull * nc + nnz;
            uint64_t frames_per_cu = std::clamp<uint64_t>((224ull << 20) / (256 * per_frame), 1, 8);
            if (const char *e = std::getenv("LDPC_AMD_FRAMES_PER_CU"))
                frames_per_cu = std::clamp<uint64_t>(std::strtoull(e, nullptr, 10), 1, 8);
            const uint32_t occ_lds = frames_per_cu >= 8 ? 0 : static_cast<uint32_t>((160 * 1024) / (frames_per_cu + 1) + 1024) & ~15u;
            check(launch_decode_mem(a, p.min_sum, plan_.max_cn_degree, occ_lds, s), "decode (memory-resident)");
        }
        else
            throw std::runtime_error("code not supported by any decoder instantiation");
    };
    // Sum-product with early termination runs in likelihood-ratio form (detmath.h: no exp/log inside the
    // iteration); the few frames whose values leave the box that form can represent come back in a list and are
    // decoded from scratch by the LLR-domain form.  Which form finishes a frame depends on that frame's data
    // only, never on the batch it travels in.  (LDPC_AMD_NO_RATIO: experiments only — results change by ulps.)
    // (codes with a check node wider than kMaxCnDegree run the LLR-domain form only; the oracle applies the same rule)
    // Without early termination the LDS-resident decoder still starts every frame in the ratio form and hands it over
    // to the LLR-domain form at an iteration boundary when its totals near the edge of the box (detmath.h "Hand-over").
    const bool handover = !p.early_term && plan_.lds_ok;
    bool later_stages = true;
    if (!p.min_sum && (p.early_term || handover) && p.iterations > 0 && plan_.max_cn_degree <= kMaxCnDegree &&
        !std::getenv("LDPC_AMD_NO_RATIO"))
    {
        uint32_t *redo = static_cast<uint32_t *>(redo_.reserve(4 * (2 * n + 1)));
        check(hipMemsetAsync(redo, 0, 4, s), "redo count");
        a.redo_count = redo, a.redo_list = redo + 1;
        if (handover)
        {
            a.redo_iter = redo + 1 + n;
            a.ws_handover = static_cast<double *>(ws_msg_.reserve(8 * n * nnz));
        }
#ifdef LDPC_AMD_PHASE_TRACE
        uint64_t *tr = nullptr;
        if (std::getenv("LDPC_AMD_PHASE_TRACE")) // debug build: per-wave phase timers of the first 2048 frames -> file
        {
            check(hipMalloc(&tr, 2048 * 32 * 8), "trace");
            check(hipMemset(tr, 0, 2048 * 32 * 8), "trace");
            a.phase_trace = tr;
        }
#endif
        launch();
#ifdef LDPC_AMD_PHASE_TRACE
        if (tr)
        {
            std::vector<uint64_t> h(2048 * 32);
            check(hipDeviceSynchronize(), "sync");
            check(hipMemcpy(h.data(), tr, 2048 * 32 * 8, hipMemcpyDeviceToHost), "trace");
            if (FILE *f = std::fopen(std::getenv("LDPC_AMD_PHASE_TRACE"), "wb"))
            {
                std::fwrite(h.data(), 8, h.size(), f);
                std::fclose(f);
            }
            (void)hipFree(tr);
            a.phase_trace = nullptr;
        }
#endif
        a.redo_count = nullptr, a.redo_list = nullptr, a.redo_iter = nullptr;
        a.redo_count_in = redo, a.redo_list_in = redo + 1;
        // A handful of frames for a caller who waits for host results anyway (the reference's decode(): one frame per call):
        // the results of the first launch and the number of frames it handed back travel to the host together; the later
        // launches — two more dispatches and a memset, a fifth of such a call's latency, for lists that are almost always
        // empty — are issued only when that number is not zero, and the results delivered again.
        const size_t pinned = st.pinned_bytes();
        if (n <= 16 && pinned && pinned <= OutStage::kPinnedLimit && !handover && a.mode == kModeLlr &&
            st.flush(s, &pin_out_, redo) == 0)
        {
            later_stages = false;
            st.items.clear(); // (delivered)
        }
        if (finished_in_one)
            later_stages = false; // (nothing has been delivered yet: the outputs are flushed below as after any last launch)
        if (!later_stages)
            ;
        else if (handover)
        {
            a.redo_iter_in = redo + 1 + n;
            a.handover_llr = fused_handover_used ? 1 : 0;
        }
        else if (plan_.lds_ok || shared6_)
        {
            // (codes the LDS-resident decoder does not take: the same three launches when the code has check nodes of degree 6,
            // which share reciprocals in the first launch of the register- and memory-resident decoders — detmath.h, dm_cn6_shared)
            // The first launch ran the shared-reciprocal check nodes (detmath.h), whose denominator products leave their
            // range in a few frames per ten thousand at the waterfall (strongly converged frames): those are decoded again
            // from scratch with every output divided separately — the ratio form still, a twentieth of a millisecond for a
            // few dozen frames — and only what leaves the box there goes on to the LLR domain.  (A lone frame takes 0.3 ms
            // in the LLR domain, and the launches of a batch run one after the other.)
            if (plan_.lds_ok)
            {
                // LDS-resident: ONE more launch, over the list — separately divided outputs and, for what leaves the box there,
                // the LLR domain, frame by frame in the same workgroup (kernels.hip, decode_kernel_list)
                a.ratio_separate = 1;
                launch();
                a.ratio_separate = 0;
                later_stages = false;
            }
            else
            {
                uint32_t *redo2 = static_cast<uint32_t *>(redo2_.reserve(4 * (n + 1)));
                check(hipMemsetAsync(redo2, 0, 4, s), "redo count");
                a.redo_count = redo2, a.redo_list = redo2 + 1;
                a.ratio_separate = 1;
                launch();
                a.ratio_separate = 0;
                a.redo_count = nullptr, a.redo_list = nullptr;
                a.redo_count_in = redo2, a.redo_list_in = redo2 + 1;
            }
        }
    }
    if (later_stages)
        launch();
    a.redo_count_in = nullptr, a.redo_list_in = nullptr, a.redo_iter_in = nullptr, a.ws_handover = nullptr, a.handover_llr = 0;
    prof_mark(0, s);
    if (a.mode == kModeAwgn && a.pairs_buffer >= 0 && ev_pairs_free_[a.pairs_buffer])
    {
        check(hipEventRecord(static_cast<hipEvent_t>(ev_pairs_free_[a.pairs_buffer]), s), "event");
        pairs_in_use_[a.pairs_buffer] = true;
    }
    if (out.codeword)
    {
        if (a.codeword)
            check(hipMemcpyAsync(out.codeword, a.codeword, n * nc,
                                 is_device_ptr(out.codeword) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s),
                  "codeword out");
        else if (is_device_ptr(out.codeword))
            check(hipMemsetAsync(out.codeword, 0, n * nc, s), "codeword out");
        else
            std::memset(out.codeword, 0, n * nc);
    }
    st.flush(s, &pin_out_);
}

void Engine::run_bec(const DecParams &p, const BatchOut &out, uint64_t n, const uint8_t *codeword, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nc = plan_.nc;
    const uint64_t nct = static_cast<uint64_t>(plan_.nct);
    const size_t state = static_cast<size_t>((plan_.nnz + 15) / 16) * 16 + 2 * ((nc + 15) / 16) * 16;
    OutStage st;
    BecArgs a{};
    a.ws = state > 160 * 1024 ? static_cast<uint8_t *>(ws_msg_.reserve(n * state)) : nullptr; // beyond LDS: state in memory
    a.plan = dev_;
    a.iterations = p.iterations;
    a.early_term = p.early_term;
    a.deg1_compat = bec_deg1_compat;
    a.n_frames = n;
    int raw_buffer = 0;
    a.raw = noise_raw_async(raw_next_, n * nct, stream, raw_buffer);
    a.eps = x_;
    a.codeword = codeword;
    a.iters = st.route(out.iters, stage_iters_, 4 * n);
    a.bit_errors = st.route(out.bit_errors, stage_be_, 4 * n);
    a.hard = st.route(out.hard, stage_hard_, n * nc);
    a.llr_out = st.route(out.llr_out, stage_llr_out_, 8 * n * nc);
    a.llr_in_dump = st.route(out.llr_in, stage_llr_in_, 8 * n * nc);
    prof_mark(0, s);
    check(launch_bec(a, s), "bec");
    prof_mark(0, s);
    noise_raw_release(raw_buffer, stream);
    if (out.codeword)
    {
        if (codeword)
            check(hipMemcpyAsync(out.codeword, codeword, n * nc,
                                 is_device_ptr(out.codeword) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s),
                  "codeword out");
        else if (is_device_ptr(out.codeword))
            check(hipMemsetAsync(out.codeword, 0, n * nc, s), "codeword out");
        else
            std::memset(out.codeword, 0, n * nc);
    }
    st.flush(s, &pin_out_);
}

// channel.cpp:44-60 for n consecutive frames (see EncodeArgs in kernels.hpp)
const uint8_t *Engine::encode_frames(uint64_t n, bool want_codewords, void *stream)
{
    if (!code_->has_G() || n == 0)
        return nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nc = plan_.nc;
    const uint64_t kc = static_cast<uint64_t>(code_->kc());
    if (code_->kc() <= 0 || code_->G.rows > code_->kc())
        throw std::runtime_error("generator matrix does not match the code (rows > nc - mc)");
    uint8_t *prev = static_cast<uint8_t *>(cw_run_.reserve(nc));
    if (!cw_run_valid_)
    {
        check(hipMemsetAsync(prev, 0, nc, s), "codeword reset");
        cw_run_valid_ = true;
    }
    check(hipMemcpyAsync(cw_before_.reserve(nc), prev, nc, hipMemcpyDeviceToDevice, s), "codeword snapshot");
    last_enc_n_ = want_codewords ? n : 0;
    EncodeArgs e{};
    e.nc = static_cast<int>(nc);
    e.kc = static_cast<int>(kc);
    e.words = static_cast<int>((kc + 63) / 64);
    e.g_col_ptr = g_col_ptr_;
    e.g_col_row = g_col_row_;
    e.g_mask = g_mask_;
    e.g_cols = code_->G.cols;
    e.info_raw = info_.generate(info_pos_, n * kc, stream);
    e.prefix = static_cast<uint64_t *>(enc_prefix_.reserve(8 * n * e.words));
    e.cw_prev = prev;
    e.codeword = want_codewords ? static_cast<uint8_t *>(cw_frames_.reserve(n * nc)) : nullptr;
    e.cw_last = static_cast<uint8_t *>(cw_next_.reserve(nc));
    e.n_frames = n;
    check(launch_encode(e, s), "encode");
    check(hipMemcpyAsync(prev, e.cw_last, nc, hipMemcpyDeviceToDevice, s), "codeword carry");
    info_pos_ += n * kc;
    return e.codeword;
}

const uint8_t *Engine::encode_frames_sharded(Comm &comm, uint64_t before, uint64_t n, uint64_t step_frames, void *stream)
{
    if (!code_->has_G() || step_frames == 0)
        return nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nc = plan_.nc;
    const uint64_t kc = static_cast<uint64_t>(code_->kc());
    if (code_->kc() <= 0 || code_->G.rows > code_->kc())
        throw std::runtime_error("generator matrix does not match the code (rows > nc - mc)");
    const int words = static_cast<int>((kc + 63) / 64);
    if (static_cast<size_t>(words) * 8 > Comm::kMaxBytes)
        throw std::runtime_error("sharded encoding: more than 2048 information bits per frame do not fit the exchange");
    uint8_t *prev = static_cast<uint8_t *>(cw_run_.reserve(nc));
    if (!cw_run_valid_)
    {
        check(hipMemsetAsync(prev, 0, nc, s), "codeword reset");
        cw_run_valid_ = true;
    }
    check(hipMemcpyAsync(cw_before_.reserve(nc), prev, nc, hipMemcpyDeviceToDevice, s), "codeword snapshot");
    last_enc_n_ = n;
    // this rank's frames: info words and their running XOR (the last entry is the XOR over the rank's range)
    uint64_t *base = static_cast<uint64_t *>(enc_base_.reserve(8 * 2 * static_cast<size_t>(words)));
    EncodeArgs e{};
    e.nc = static_cast<int>(nc), e.kc = static_cast<int>(kc), e.words = words;
    e.g_col_ptr = g_col_ptr_, e.g_col_row = g_col_row_, e.g_cols = code_->G.cols;
    e.g_mask = g_mask_;
    e.cw_prev = prev;
    e.cw_last = static_cast<uint8_t *>(cw_next_.reserve(nc));
    std::vector<uint64_t> mine(words, 0), all(static_cast<size_t>(words) * comm.world());
    if (n)
    {
        e.info_raw = info_.generate(info_pos_ + before * kc, n * kc, stream);
        e.prefix = static_cast<uint64_t *>(enc_prefix_.reserve(8 * n * words));
        e.n_frames = n;
        check(launch_encode_prefix(e, s), "encode (info words)");
        check(hipMemcpyAsync(mine.data(), e.prefix + (n - 1) * words, 8 * static_cast<size_t>(words), hipMemcpyDeviceToHost, s), "info sum");
        check(hipStreamSynchronize(s), "sync");
    }
    // ONE exchange: every rank's sum; the ranks before this one give the word every prefix of this rank starts from, all of
    // them together the codeword after the step (the codeword accumulates linearly: channel.cpp:44-60)
    const auto t0 = std::chrono::steady_clock::now();
    comm.all_gather(mine.data(), all.data(), 8 * static_cast<size_t>(words));
    host_ms_[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<uint64_t> host(2 * static_cast<size_t>(words), 0); // [0, words): ranks before this one; [words, 2 words): all
    for (int q = 0; q < comm.world(); ++q)
        for (int w = 0; w < words; ++w)
        {
            if (q < comm.rank())
                host[w] ^= all[static_cast<size_t>(q) * words + w];
            host[words + w] ^= all[static_cast<size_t>(q) * words + w];
        }
    check(hipMemcpyAsync(base, host.data(), 8 * host.size(), hipMemcpyHostToDevice, s), "info sums");
    check(hipStreamSynchronize(s), "sync"); // (host is a local: the copy must have read it)
    const uint8_t *cw = nullptr;
    if (n)
    {
        e.base = base;
        e.codeword = static_cast<uint8_t *>(cw_frames_.reserve(n * nc));
        check(launch_encode_codewords(e, s, false), "encode (codewords)");
        cw = e.codeword;
    }
    // the codeword after the step, on every rank: cw_prev ^ (sum over all ranks) G — one pseudo-frame whose prefix is that sum
    e.prefix = base + words, e.base = nullptr, e.n_frames = 1, e.codeword = nullptr;
    check(launch_encode_codewords(e, s, true), "encode (codeword carry)");
    check(hipMemcpyAsync(prev, e.cw_last, nc, hipMemcpyDeviceToDevice, s), "codeword carry");
    info_pos_ += step_frames * kc;
    return cw;
}

void Engine::decode_llr(const DecParams &p, uint64_t n, const double *llr_in, const BatchOut &out, void *stream)
{
    if (n == 0)
        return;
    upload_plan();
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t nc = static_cast<uint64_t>(plan_.nc), sub = max_sub_batch();
    for (uint64_t done = 0; done < n; done += sub)
    {
        const uint64_t m = std::min(sub, n - done);
        const size_t bytes = 8 * m * nc;
        BatchOut o = out;
        if (o.iters) o.iters += done;
        if (o.bit_errors) o.bit_errors += done;
        if (o.hard) o.hard += done * nc;
        if (o.llr_out) o.llr_out += done * nc;
        if (o.llr_in) o.llr_in += done * nc;
        if (o.codeword) o.codeword += done * nc;
        DecodeArgs a{};
        a.mode = kModeLlr;
        const double *src = llr_in + done * nc;
        if (is_device_ptr(src))
            a.llr_in = src;
        else
        {
            void *d = stage_in_.reserve(bytes);
            const void *from = src;
            if (bytes <= (1u << 20)) // small inputs through page-locked memory (see OutStage::flush)
            {
                if (!pin_in_ev_)
                {
                    hipEvent_t ev;
                    check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
                    pin_in_ev_ = ev;
                }
                else
                    check(hipEventSynchronize(static_cast<hipEvent_t>(pin_in_ev_)), "event"); // the previous copy out of the buffer is done
                void *h = pin_in_.reserve(bytes);
                std::memcpy(h, src, bytes);
                from = h;
            }
            check(hipMemcpyAsync(d, from, bytes, hipMemcpyHostToDevice, s), "copy in");
            if (from != src)
                check(hipEventRecord(static_cast<hipEvent_t>(pin_in_ev_), s), "event");
            a.llr_in = static_cast<const double *>(d);
        }
        run_decode(a, p, o, m, stream);
    }
}

void Engine::stream_rewind_encoder(uint64_t frames_back, void *stream)
{
    if (!code_->has_G() || frames_back == 0)
        return;
    if (frames_back > last_enc_n_)
        throw std::runtime_error("stream_rewind_encoder: more frames than the last batch held");
    bind_device();
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nc = plan_.nc;
    const uint8_t *src = frames_back == last_enc_n_
                             ? static_cast<const uint8_t *>(cw_before_.get())
                             : static_cast<const uint8_t *>(cw_frames_.get()) + (last_enc_n_ - frames_back - 1) * nc;
    check(hipMemcpyAsync(cw_run_.get(), src, nc, hipMemcpyDeviceToDevice, s), "codeword rewind");
    info_pos_ -= frames_back * static_cast<uint64_t>(code_->kc());
    last_enc_n_ = 0;
}

// ---------------------------------------------------------------------------------------------
void Engine::stream_begin(int channel, uint64_t seed, double x, bool fresh)
{
    if (channel != kAwgn && channel != kBsc && channel != kBec)
        throw std::runtime_error("No channel selected.");
    chan_ = channel;
    // jump-ahead in four thread groups where its latency counts: the erasure channel (its bit-sliced decoder leaves the noise
    // chain on the critical path) and beside the register-resident decoders (rng_kernels.hip, mt_jump_kernel)
    noise_.st.set_jump_groups(channel == kBec || reg_plan_.ok ? 4 : 1);
    x_ = x;
    frame_pos_ = 0;
    raw_next_ = 0;
    cur_pair_ = cur_chunk_ = cur_k_ = 0;
    small_cache_.valid = false;
    sh_chunk_ = sh_pairs_ = 0;
    stream_mode_ = 0;
    noise_.reset(seed);
    if (fresh || info_.seed() != (seed << 1))
    {
        info_.reset(seed << 1); // channel.cpp:11
        info_pos_ = 0;
        cw_run_valid_ = false;
    }
    if (channel == kAwgn)
    {
        sigma2_ = std::pow(10, -x / 10); // channel.cpp:39
        sigma_ = std::sqrt(sigma2_);
    }
    else
        delta_ = std::log((1 - x) / x); // channel.cpp:139
}

void Engine::ensure_rng_stream()
{
    if (rng_stream_)
        return;
    // non-blocking: no implicit ordering with the caller's (possibly default) stream
    hipStream_t rs;
    check(hipStreamCreateWithFlags(&rs, hipStreamNonBlocking), "hipStreamCreate");
    rng_stream_ = rs;
    for (int i = 0; i < 2; ++i)
    {
        hipEvent_t e0, e1;
        check(hipEventCreateWithFlags(&e0, hipEventDisableTiming), "hipEventCreate");
        check(hipEventCreateWithFlags(&e1, hipEventDisableTiming), "hipEventCreate");
        ev_pairs_ready_[i] = e0, ev_pairs_free_[i] = e1;
    }
}

const uint64_t *Engine::noise_raw_async(uint64_t first, uint64_t count, void *stream, int &buffer)
{
    ensure_rng_stream();
    hipStream_t s = static_cast<hipStream_t>(rng_stream_), user = static_cast<hipStream_t>(stream);
    const int buf = pp_;
    pp_ ^= 1;
    // the launch that last read this buffer (two batches ago) must be done before it is refilled
    if (pairs_in_use_[buf])
        check(hipStreamWaitEvent(s, static_cast<hipEvent_t>(ev_pairs_free_[buf]), 0), "wait raw free");
    prof_mark(1, s);
    const uint64_t *raw = noise_.generate(first, count, s, buf);
    prof_mark(1, s);
    check(hipEventRecord(static_cast<hipEvent_t>(ev_pairs_ready_[buf]), s), "event");
    check(hipStreamWaitEvent(user, static_cast<hipEvent_t>(ev_pairs_ready_[buf]), 0), "wait raw ready");
    buffer = buf;
    return raw;
}

void Engine::noise_raw_release(int buffer, void *stream)
{
    check(hipEventRecord(static_cast<hipEvent_t>(ev_pairs_free_[buffer]), static_cast<hipStream_t>(stream)), "event");
    pairs_in_use_[buffer] = true;
}

// One pass of the noise generator on the side stream: `full` whole chunks from `chunk` on (+ a prefix of last_blocks twist
// blocks of the chunk after them) -> normals in the slabs of buffer `buf`, their counts, the slab table, and where the
// consumer stands afterwards.  Chunk start states: the ring (one rank reading front to back) or the strided table (sharded).
Engine::NoisePass Engine::noise_pass(uint64_t chunk, uint32_t full, uint32_t last_blocks, uint32_t n_piece, uint64_t need, uint64_t target,
                                     int buf, bool write_normals, bool strided, uint64_t stride)
{
    hipStream_t s = static_cast<hipStream_t>(rng_stream_);
    MtDevice &st = noise_.st;
    NoisePass np;
    np.n_slabs = full + (last_blocks ? 1 : 0);
    if (np.n_slabs == 0)
        throw std::runtime_error("noise generator: empty pass");
    if (np.n_slabs > (strided ? StridedTable::kMaxRows : StateRing::kWindow))
        throw std::runtime_error("noise generator: batch spans more chunks than the state table holds");
    const uint64_t slab_words = 2 * st.chunk_trials();
    NormalsArgs na{};
    if (strided)
    {
        na.ring = st.ensure_strided(chunk, np.n_slabs, stride, s);
        na.ring_rows = MtDevice::strided_rows();
        na.first_row = 0;
    }
    else
    {
        na.first_row = st.ensure_ring(chunk, chunk + np.n_slabs, s);
        na.ring = st.ring();
        na.ring_rows = MtDevice::ring_rows();
    }
    // (two slabs of slack: the number of chunks a batch spans varies by one, and growing a buffer means freeing it first)
    np.slabs = write_normals ? static_cast<uint64_t *>(slabs_[buf].reserve(8 * slab_words * (np.n_slabs + 2))) : nullptr;
    // (a counting pass must not touch the slab table a decode launch in flight may still read)
    np.cum = static_cast<uint64_t *>((write_normals ? slab_cum_[buf] : nz_cum_skip_).reserve(8 * (static_cast<size_t>(np.n_slabs) + 8)));
    uint32_t *counts = static_cast<uint32_t *>(nz_counts_.reserve(4 * static_cast<size_t>(np.n_slabs)));
    NormalsResult *res = static_cast<NormalsResult *>(nz_result_.reserve(sizeof(NormalsResult)));
    na.slabs = np.slabs;
    na.slab_words = slab_words;
    na.counts = counts;
    na.write_normals = write_normals ? 1 : 0;
    na.locate_chunk = 0xFFFFFFFFu;
    na.pack = noise_.pack();
    na.raw = static_cast<uint64_t *>(nz_raw_.reserve(8 * st.chunk_words() * (np.n_slabs + 2)));
    na.lookback = static_cast<uint64_t *>(nz_lookback_.reserve(8 * normals_lookback_words(np.n_slabs + 2, st.chunk_blocks())));
    na.n_chunks = np.n_slabs;
    na.blocks = st.chunk_blocks();
    na.last_blocks = last_blocks ? last_blocks : st.chunk_blocks(); // (the prefix chunk rides in the same two launches)
    check(launch_mt_normals(na, s), "mt_normals");
    check(launch_normals_finish(counts, np.n_slabs, n_piece, full, need, target, np.cum, res, s), "normals_finish");
    check(hipMemcpyAsync(&np.res, res, sizeof np.res, hipMemcpyDeviceToHost, s), "noise result");
    const auto t0 = std::chrono::steady_clock::now();
    check(hipStreamSynchronize(s), "sync"); // waits for the noise-stream kernels only, not for the caller's decode
    host_ms_[1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), ++host_n_[1];
    return np;
}

void Engine::fill_slab_args(DecodeArgs &a, const NoisePass &np, uint64_t pair_origin, int buf) const
{
    a.pairs = np.slabs;
    a.slab_cum = np.cum;
    a.slab_words = 2 * noise_.st.chunk_trials();
    a.n_slabs = np.n_slabs;
    a.slab_pairs_inv = static_cast<float>(1.0 / (static_cast<double>(noise_.st.chunk_trials()) * 0.7853981633974483));
    a.pair_origin = pair_origin;
    a.sigma = sigma_, a.sigma2 = sigma2_, a.inv_sigma2 = 1.0 / sigma2_;
    a.shorten_llr = 99999.9; // channel.cpp:83
    a.pairs_buffer = buf;
}

// The normals of frames [frame_pos_, frame_pos_ + n): normal g is element g & 1 of accepted pair g >> 1.
void Engine::awgn_prepare(uint64_t n, DecodeArgs &a, void *stream, bool write_normals)
{
    hipStream_t user = static_cast<hipStream_t>(stream);
    ensure_rng_stream();
    hipStream_t s = static_cast<hipStream_t>(rng_stream_);
    const uint64_t nct = static_cast<uint64_t>(plan_.nct);
    const uint64_t g0 = frame_pos_ * nct, g1 = g0 + n * nct; // normals [g0, g1)
    const uint64_t q0 = g0 >> 1, q1 = (g1 - 1) >> 1, qn = g1 >> 1;
    if (q0 != cur_pair_)
        throw std::runtime_error("noise stream out of step");
    // pairs that must exist from the start of chunk cur_chunk_ on / list index of the first pair the NEXT batch needs
    const uint64_t need = cur_k_ + (q1 - q0 + 1), target = cur_k_ + (qn - q0);
    if (write_normals && small_cache_.valid && small_cache_.chunk == cur_chunk_ && need <= small_cache_.np.res.total &&
        target < small_cache_.np.res.total && !std::getenv("LDPC_AMD_NO_SMALL_CACHE"))
    {
        // (the slab was complete when the pass that made it returned, and the decode launches that read it since are
        // ordered on the caller's stream by the events of run_decode as for any batch)
        check(hipStreamWaitEvent(user, static_cast<hipEvent_t>(ev_pairs_ready_[small_cache_.buf]), 0), "wait pairs ready");
        fill_slab_args(a, small_cache_.np, q0 - cur_k_, small_cache_.buf);
        a.normal_base = g0;
        cur_k_ = target;
        cur_pair_ = qn;
        return;
    }
    const int buf = pp_;
    if (write_normals)
    {
        pp_ ^= 1;
        if (small_cache_.buf == buf)
            small_cache_.valid = false;
        // the decode kernel that last read this slab buffer (two batches ago) must be done before it is refilled
        if (pairs_in_use_[buf])
            check(hipStreamWaitEvent(s, static_cast<hipEvent_t>(ev_pairs_free_[buf]), 0), "wait pairs free");
    }
    const uint64_t ct = noise_.st.chunk_trials();
    double trials = static_cast<double>(need) * 1.2732395447351628 + 8.0 * std::sqrt(static_cast<double>(need)) + 256;
    // a small request: two dozen frames further than asked (a fraction of the prefix's serial chain), for the requests
    // that are likely to follow (small_cache_)
    const bool small = write_normals && n <= 16;
    if (small)
        trials += 24.0 * static_cast<double>(nct / 2 + 1) * 1.2732395447351628;
    NoisePass np;
    PhaseTrace tr;
    prof_mark(1, s);
    for (;;)
    {
        const uint64_t t = static_cast<uint64_t>(trials);
        uint64_t full = t / ct;
        uint32_t last_blocks = static_cast<uint32_t>((t - full * ct + kBlockTrials - 1) / kBlockTrials);
        if (last_blocks >= noise_.st.chunk_blocks())
            ++full, last_blocks = 0;
        np = noise_pass(cur_chunk_, static_cast<uint32_t>(full), last_blocks, static_cast<uint32_t>(full + (last_blocks ? 1 : 0)), need, target,
                        buf, write_normals, false, 0);
        tr.mark("noise pass");
        if (np.res.enough)
            break;
        trials += trials / 8 + 4096; // vanishingly rare: take a longer look at the same stream
    }
    prof_mark(1, s);
    if (write_normals)
    {
        check(hipEventRecord(static_cast<hipEvent_t>(ev_pairs_ready_[buf]), s), "event");
        check(hipStreamWaitEvent(user, static_cast<hipEvent_t>(ev_pairs_ready_[buf]), 0), "wait pairs ready");
        fill_slab_args(a, np, q0 - cur_k_, buf);
        a.normal_base = g0;
    }
    if (write_normals)
    {
        small_cache_.valid = small && np.n_slabs == 1 && np.res.next_slab == 0;
        small_cache_.chunk = cur_chunk_, small_cache_.buf = buf, small_cache_.np = np;
    }
    cur_chunk_ += np.res.next_slab;
    cur_k_ = np.res.next_k;
    cur_pair_ = qn;
    // the chunk start states of the next batch but one, now: their jump-ahead chain then runs beside the next batch's
    // generator chain (both hide under decode kernels whose waves outrank them, kernels.hip), not in front of it
    noise_.st.prefetch_ring(cur_chunk_ + 2 * (static_cast<uint64_t>(np.n_slabs) + 2));
}

// raw 64-bit draws consumed from the noise stream since stream_begin (what orc_chan_raw_draws counts)
uint64_t Engine::stream_raw_draws()
{
    if (chan_ != kAwgn)
        return raw_next_;
    if (stream_mode_ == 2)
        return sh_chunk_ * noise_.st.chunk_words();
    const uint64_t nct = static_cast<uint64_t>(plan_.nct);
    const uint64_t g1 = frame_pos_ * nct;
    if (g1 == 0)
        return 0;
    // the last pair the stream has drawn: (g1 - 1) >> 1.  cur_* describe pair g1 >> 1: the same pair (its second normal is
    // still saved) when g1 is odd, the pair before it otherwise — which may be the last accepted pair of the previous chunk.
    bind_device();
    ensure_rng_stream();
    hipStream_t s = static_cast<hipStream_t>(rng_stream_);
    uint64_t chunk = cur_chunk_;
    uint32_t rank = static_cast<uint32_t>(cur_k_);
    if ((g1 & 1) == 0)
    {
        if (cur_k_ > 0)
            --rank;
        else
            --chunk, rank = 0xFFFFFFFFu;
    }
    MtDevice &st = noise_.st;
    NormalsArgs na{};
    na.first_row = st.ensure_ring(chunk, chunk + 1, s);
    na.ring = st.ring();
    na.ring_rows = MtDevice::ring_rows();
    na.n_chunks = 1;
    na.blocks = na.last_blocks = st.chunk_blocks();
    na.slab_words = 2 * st.chunk_trials();
    na.counts = static_cast<uint32_t *>(nz_counts_.reserve(4));
    na.write_normals = 0;
    na.locate_chunk = 0;
    na.locate_rank = rank;
    na.locate_out = static_cast<uint64_t *>(nz_locate_.reserve(8));
    na.pack = 1;
    na.raw = static_cast<uint64_t *>(nz_raw_.reserve(8 * st.chunk_words()));
    na.lookback = static_cast<uint64_t *>(nz_lookback_.reserve(8 * normals_lookback_words(1, st.chunk_blocks())));
    const uint64_t none = rank == 0xFFFFFFFFu ? 0 : ~0ull; // ("last pair": a running maximum over the chunk's workgroups)
    uint64_t t = none;
    check(hipMemcpyAsync(na.locate_out, &t, 8, hipMemcpyHostToDevice, s), "locate");
    check(hipStreamSynchronize(s), "sync");
    check(launch_mt_normals(na, s), "mt_normals (locate)");
    check(hipMemcpyAsync(&t, na.locate_out, 8, hipMemcpyDeviceToHost, s), "locate");
    check(hipStreamSynchronize(s), "sync");
    if (rank != 0xFFFFFFFFu && t == none)
        throw std::runtime_error("noise stream position not found");
    return 2 * (chunk * st.chunk_trials() + t + 1);
}

void Engine::stream_skip(uint64_t n_frames, void *stream)
{
    if (!chan_)
        throw std::runtime_error("stream_begin() has not been called");
    if (stream_mode_ == 2)
        throw std::runtime_error("stream_skip after stream_decode_sharded on the same stream: call stream_begin first");
    stream_mode_ = 1;
    upload_plan();
    const uint64_t nct = static_cast<uint64_t>(plan_.nct);
    while (n_frames)
    {
        const uint64_t n = std::min<uint64_t>(n_frames, max_sub_batch());
        encode_frames(n, false, stream);
        if (chan_ == kAwgn)
        {
            DecodeArgs a{};
            awgn_prepare(n, a, stream, /*write_normals=*/false);
        }
        else
            raw_next_ += n * nct;
        frame_pos_ += n;
        n_frames -= n;
    }
}

void Engine::stream_decode(const DecParams &p, uint64_t n_frames, const BatchOut &out, void *stream)
{
    if (!chan_)
        throw std::runtime_error("stream_begin() has not been called");
    if (n_frames == 0)
        return;
    if (stream_mode_ == 2)
        throw std::runtime_error("stream_decode after stream_decode_sharded on the same stream: call stream_begin first");
    stream_mode_ = 1;
    upload_plan();
    const uint64_t nct = static_cast<uint64_t>(plan_.nct), nc = static_cast<uint64_t>(plan_.nc);
    const uint64_t sub = max_sub_batch();
    uint64_t done = 0;
    while (done < n_frames)
    {
        const uint64_t n = std::min<uint64_t>(n_frames - done, sub);
        BatchOut o = out;
        if (o.iters) o.iters += done;
        if (o.bit_errors) o.bit_errors += done;
        if (o.hard) o.hard += done * nc;
        if (o.llr_out) o.llr_out += done * nc;
        if (o.llr_in) o.llr_in += done * nc;
        if (o.codeword) o.codeword += done * nc;
        PhaseTrace tr;
        const uint8_t *cw = encode_frames(n, true, stream);
        DecodeArgs a{};
        a.codeword = cw;
        if (chan_ == kAwgn)
        {
            a.mode = kModeAwgn;
            awgn_prepare(n, a, stream);
            tr.mark("awgn_prepare");
            run_decode(a, p, o, n, stream);
            tr.mark("run_decode(enq)");
        }
        else if (chan_ == kBsc)
        {
            a.mode = kModeBsc;
            int raw_buffer = 0;
            a.raw = noise_raw_async(raw_next_, n * nct, stream, raw_buffer);
            a.eps = x_, a.delta = delta_;
            a.shorten_llr = delta_; // channel.cpp:152
            run_decode(a, p, o, n, stream);
            noise_raw_release(raw_buffer, stream);
            raw_next_ += n * nct;
        }
        else
        {
            run_bec(p, o, n, cw, stream);
            raw_next_ += n * nct;
        }
        frame_pos_ += n;
        done += n;
    }
}

// ---------------------------------------------------------------------------------------------
// Geometry of a sharded AWGN step of about target_frames frames over `world` ranks: every rank's piece is `m` whole
// generator chunks (so that its chunk start states are the previous step's advanced by ONE polynomial, world * m chunks),
// followed by a margin of one frame's worth of trials from the next chunk.
namespace
{
struct ShardGeometry
{
    uint32_t m;             // chunks per piece
    uint32_t margin_blocks; // twist blocks of the chunk after the piece that are generated as well
};
ShardGeometry shard_geometry(uint64_t target_frames, int world, uint64_t nct, uint64_t chunk_trials, uint32_t chunk_blocks)
{
    const double pairs_per_rank = static_cast<double>(std::max<uint64_t>(target_frames, 1)) * static_cast<double>(nct) / 2.0 / world;
    const double chunks = pairs_per_rank * 1.2732395447351628 / static_cast<double>(chunk_trials);
    ShardGeometry g;
    g.m = static_cast<uint32_t>(std::clamp<double>(std::floor(chunks + 0.5), 1.0, static_cast<double>(StridedTable::kMaxRows - 1)));
    // the pairs of one frame beyond the piece (the last frame a rank owns may end in its neighbour's piece): twice the
    // expected trials plus 8192, in whole blocks
    const uint64_t margin_trials = (nct / 2 + 2) * 2 + 8192;
    g.margin_blocks = static_cast<uint32_t>(std::min<uint64_t>((margin_trials + kBlockTrials - 1) / kBlockTrials, chunk_blocks));
    return g;
}
} // namespace

uint64_t Engine::shard_capacity(uint64_t target_frames, int world) const
{
    world = std::max(world, 1);
    const uint64_t nct = static_cast<uint64_t>(plan_.nct);
    const uint64_t even = (std::max<uint64_t>(target_frames, 1) + world - 1) / world; // BSC / BEC: even split
    // AWGN: a frame belongs to the rank whose piece holds its first pair; a piece of m chunks holds at most m * chunk_trials
    // pairs (every trial accepted), i.e. at most that many / (nct / 2) frame starts — a bound, not a statistical estimate
    const ShardGeometry g = shard_geometry(target_frames, world, nct, noise_.st.chunk_trials(), noise_.st.chunk_blocks());
    const uint64_t awgn = (2 * static_cast<uint64_t>(g.m) * noise_.st.chunk_trials() + nct - 1) / nct + 2;
    return std::max(even, awgn);
}

void Engine::encoder_snapshot(void *stream)
{
    if (!code_->has_G())
        return;
    bind_device();
    const size_t nc = plan_.nc;
    enc_snap_pos_ = info_pos_;
    enc_snap_valid_ = cw_run_valid_;
    if (cw_run_valid_)
        check(hipMemcpyAsync(enc_snap_.reserve(nc), cw_run_.get(), nc, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)),
              "encoder snapshot");
}

void Engine::encoder_restore_and_skip(uint64_t frames, void *stream)
{
    if (!code_->has_G())
        return;
    bind_device();
    const size_t nc = plan_.nc;
    info_pos_ = enc_snap_pos_;
    cw_run_valid_ = enc_snap_valid_;
    if (enc_snap_valid_)
        check(hipMemcpyAsync(cw_run_.reserve(nc), enc_snap_.get(), nc, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)),
              "encoder restore");
    for (uint64_t left = frames; left;)
    {
        const uint64_t n = std::min<uint64_t>(left, 1u << 17);
        encode_frames(n, false, stream);
        left -= n;
    }
    last_enc_n_ = 0;
}

Engine::ShardStep Engine::stream_decode_sharded(Comm &comm, const DecParams &p, uint64_t target_frames, const BatchOut &out,
                                                void *stream, const std::string *failed_before)
{
    // (argument errors: the same on every rank, thrown before anything is exchanged)
    if (!chan_)
        throw std::runtime_error("stream_begin() has not been called");
    const int R = comm.world(), r = comm.rank();
    const uint64_t nct = static_cast<uint64_t>(plan_.nct), cap = shard_capacity(target_frames, R);
    target_frames = std::max<uint64_t>(target_frames, 1);
    if (stream_mode_ == 1)
        throw std::runtime_error("stream_decode_sharded after stream_decode on the same stream: call stream_begin first");
    stream_mode_ = 2;
    if (chan_ != kAwgn && !failed_before)
        upload_plan();
    ShardStep st;
    st.step_first = frame_pos_;
    DecodeArgs a{};
    if (chan_ == kAwgn)
    {
        // The step: world * m whole chunks of the raw stream.  Rank r generates chunks [base + r m, base + (r+1) m) and the
        // head of the next one, counts its accepted pairs, and ONE all-gather of {pairs in the piece, pairs incl. the
        // margin, status} places every piece in the pair sequence.  Every rank can then evaluate every rank's conditions:
        // a failure anywhere makes all ranks throw together instead of leaving the others in the next collective.
        const ShardGeometry g = shard_geometry(target_frames, R, nct, noise_.st.chunk_trials(), noise_.st.chunk_blocks());
        const uint64_t stride = static_cast<uint64_t>(R) * g.m;
        uint64_t send[3] = {0, 0, 0};
        NoisePass np;
        std::string local_error;
        int buf = -1;
        try
        {
            // (a caller whose own preparation failed on this rank — the simulation loop's encoder snapshot — still has to
            // take part in the exchange below, or the other ranks wait in it for ever: it enters with the failure in hand)
            if (failed_before)
                throw std::runtime_error(*failed_before);
            if (cap > max_sub_batch())
                throw std::runtime_error("sharded step too large for one launch per rank");
            upload_plan(); // (the first touch of the GPU: a rank without a usable device fails here, inside the guarded part)
            ensure_rng_stream();
            hipStream_t s = static_cast<hipStream_t>(rng_stream_);
            buf = pp_;
            pp_ ^= 1;
            if (small_cache_.buf == buf)
                small_cache_.valid = false;
            if (pairs_in_use_[buf])
                check(hipStreamWaitEvent(s, static_cast<hipEvent_t>(ev_pairs_free_[buf]), 0), "wait pairs free");
            prof_mark(1, s);
            np = noise_pass(sh_chunk_ + static_cast<uint64_t>(r) * g.m, g.m, g.margin_blocks, g.m, 0, 0, buf, true, true, stride);
            prof_mark(1, s);
            send[0] = np.res.piece, send[1] = np.res.total;
            // the chunk start states of the next step but one, now: their jump-ahead chain has a whole step to run, on a stream
            // of its own, and every step's generator chain starts at once (MtDevice::prefetch_strided)
            noise_.st.prefetch_strided();
        }
        catch (const std::exception &e)
        {
            local_error = e.what();
            send[2] = 1;
        }
        std::vector<uint64_t> all(3 * static_cast<size_t>(R));
        const auto t0 = std::chrono::steady_clock::now();
        comm.all_gather(send, all.data(), sizeof send);
        host_ms_[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), ++host_n_[0];
        const ShardPlacement pl = place_pieces(all.data(), R, r, sh_pairs_, frame_pos_, nct, cap, local_error); // (shard_place.hpp)
        st.first = pl.first, st.n = pl.n, st.step_frames = pl.step_frames;
        hipStream_t s = static_cast<hipStream_t>(rng_stream_), user = static_cast<hipStream_t>(stream);
        check(hipEventRecord(static_cast<hipEvent_t>(ev_pairs_ready_[buf]), s), "event");
        check(hipStreamWaitEvent(user, static_cast<hipEvent_t>(ev_pairs_ready_[buf]), 0), "wait pairs ready");
        a.mode = kModeAwgn;
        fill_slab_args(a, np, pl.pair_start, buf);
        a.normal_base = st.first * nct;
        sh_pairs_ = pl.pairs_after;
        sh_chunk_ += stride;
    }
    else
    {
        if (failed_before) // (no exchange inside a BSC / BEC step: the caller's own exchange carries the failure)
            throw std::runtime_error(*failed_before);
        if (cap > max_sub_batch())
            throw std::runtime_error("sharded step too large for one launch per rank");
        st.step_frames = target_frames;
        const uint64_t base = target_frames / R, extra = target_frames % R;
        st.n = base + (static_cast<uint64_t>(r) < extra ? 1 : 0);
        st.first = st.step_first + base * r + std::min<uint64_t>(r, extra);
    }
    // encoder: a rank draws the info words of its own frames only; one all-gather of the ranks' info-word sums (ceil(kc / 64)
    // words) gives each the word its prefixes start from and all of them the codeword accumulated over the step
    const uint8_t *cw = nullptr;
    if (code_->has_G())
    {
        cw = encode_frames_sharded(comm, st.first - st.step_first, st.n, st.step_frames, stream);
        last_enc_n_ = 0;
    }
    if (st.n)
    {
        if (chan_ == kAwgn)
        {
            a.codeword = cw;
            run_decode(a, p, out, st.n, stream);
        }
        else if (chan_ == kBsc)
        {
            a.mode = kModeBsc;
            a.codeword = cw;
            int raw_buffer = 0;
            a.raw = noise_raw_async(st.first * nct, st.n * nct, stream, raw_buffer);
            a.eps = x_, a.delta = delta_;
            a.shorten_llr = delta_; // channel.cpp:152
            run_decode(a, p, out, st.n, stream);
            noise_raw_release(raw_buffer, stream);
        }
        else
        {
            const uint64_t keep_raw = raw_next_;
            raw_next_ = st.first * nct; // run_bec reads the stream at raw_next_
            run_bec(p, out, st.n, cw, stream);
            raw_next_ = keep_raw;
        }
    }
    else if (chan_ == kAwgn && a.pairs_buffer >= 0)
        pairs_in_use_[a.pairs_buffer] = false;
    if (chan_ != kAwgn)
        raw_next_ = (st.step_first + st.step_frames) * nct;
    frame_pos_ = st.step_first + st.step_frames;
    return st;
}

} // namespace ldpc_amd
