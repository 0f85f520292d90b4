// engine.cpp — see engine.hpp.  Compiled by hipcc (host code using the HIP runtime API).
#include "engine.hpp"

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
    // small results go through page-locked memory: a device-to-pageable copy is a blocking staged copy per call,
    // which is most of a single-frame decode()'s latency
    void flush(hipStream_t s, PinnedBuffer *pin = nullptr)
    {
        size_t total = 0;
        for (auto &i : items)
            total += (i.bytes + 63) & ~size_t(63);
        if (pin && total && total <= kPinnedLimit)
        {
            char *h = static_cast<char *>(pin->reserve(total));
            size_t off = 0;
            for (auto &i : items)
            {
                check(hipMemcpyAsync(h + off, i.dev, i.bytes, hipMemcpyDeviceToHost, s), "copy out");
                off += (i.bytes + 63) & ~size_t(63);
            }
            check(hipStreamSynchronize(s), "sync");
            off = 0;
            for (auto &i : items)
            {
                std::memcpy(i.host, h + off, i.bytes);
                off += (i.bytes + 63) & ~size_t(63);
            }
            items.clear();
            return;
        }
        for (auto &i : items)
            check(hipMemcpyAsync(i.host, i.dev, i.bytes, hipMemcpyDeviceToHost, s), "copy out");
        if (!items.empty())
            check(hipStreamSynchronize(s), "sync");
        items.clear();
    }
    static constexpr size_t kPinnedLimit = 1 << 20;
};

// t^(J*2^m) mod phi, computed lazily once per process
const Gf2Poly &jump_poly(unsigned m)
{
    static std::mutex mu;
    static std::vector<Gf2Poly> polys;
    std::lock_guard<std::mutex> lk(mu);
    if (mt64_charpoly().empty())
        throw std::runtime_error("mt19937_64 characteristic polynomial has unexpected degree");
    while (polys.size() <= m)
    {
        if (polys.empty())
            polys.push_back(mt64_pow_t(MtStream::kChunkWords));
        else
            polys.push_back(gf2_mulmod(polys.back(), polys.back()));
    }
    return polys[m];
}
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
        size_t want = std::max(bytes, size_ + size_ / 2);
        check(hipMalloc(&ptr_, want), "hipMalloc");
        size_ = want;
    }
    return ptr_;
}

// ---------------------------------------------------------------------------------------------
void ChunkTable::ensure(uint64_t c_lo, uint64_t c_hi, std::vector<ChunkTableOp> &ops)
{
    if (c_hi - c_lo > kCap)
        throw std::runtime_error("mt19937_64 stream request exceeds the chunk-state table");
    if (!valid_ || c_lo < base_)
    {
        ops.push_back({ChunkTableOp::kUploadWindow0, 0, 0});
        base_ = 0;
        ready_ = 1;
        pow_ready_ = 1;
        valid_ = true;
    }
    auto rebase = [&](uint64_t c) {
        ops.push_back({ChunkTableOp::kRebase, static_cast<uint32_t>(c - base_), 0});
        base_ = c;
        ready_ = 1;
        pow_ready_ = 1;
    };
    for (;;)
    {
        const uint64_t need = c_hi - base_;
        if (need <= ready_)
            return;
        if (need > kCap)
        {
            if (c_lo - base_ < ready_)
            {
                rebase(c_lo);
                continue;
            }
            if (pow_ready_ >= kCap) // far seek: stride forward by the table length
            {
                rebase(base_ + ready_ - 1);
                continue;
            }
        }
        // extend by doubling: rows [pow, 2*pow) = jump_{J*pow}(rows [0, pow))
        unsigned m = 0;
        while ((1u << m) < pow_ready_)
            ++m;
        ops.push_back({ChunkTableOp::kJump, pow_ready_, m});
        pow_ready_ *= 2;
        ready_ = std::max(ready_, pow_ready_);
    }
}

void ChunkTable::note_next_written(uint64_t c_hi)
{
    if (c_hi - base_ == ready_ && ready_ <= kCap)
        ++ready_; // the state that follows the last generated chunk came for free (row index ready_ <= kCap)
}

void MtStream::reset(uint64_t seed)
{
    if (seeded_ && seed == seed_)
        return; // chunk states depend on the seed only; keep them
    seed_ = seed;
    seeded_ = true;
    table_.invalidate();
}

void MtStream::ensure_states(uint64_t c_lo, uint64_t c_hi, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t row = sizeof(uint64_t) * kMtWords;
    uint64_t *st = static_cast<uint64_t *>(states_.reserve(row * (kStateCap + 1)));
    std::vector<ChunkTableOp> ops;
    table_.ensure(c_lo, c_hi, ops);
    for (const ChunkTableOp &op : ops)
        switch (op.kind)
        {
        case ChunkTableOp::kUploadWindow0:
        {
            uint64_t w0[kMtWords];
            mt64_window0(seed_, w0);
            check(hipMemcpyAsync(st, w0, row, hipMemcpyHostToDevice, s), "upload window0");
            check(hipStreamSynchronize(s), "sync"); // w0 is a stack buffer
            break;
        }
        case ChunkTableOp::kRebase:
            check(hipMemcpyAsync(st, st + static_cast<size_t>(op.a) * kMtWords, row, hipMemcpyDeviceToDevice, s), "rebase");
            break;
        case ChunkTableOp::kJump:
            check(launch_mt_jump(st, st + static_cast<size_t>(op.a) * kMtWords, device_poly(op.b, stream), op.a, s), "mt_jump");
            break;
        }
}

// device copy of t^(J*2^m) mod phi, uploaded once per stream object
const uint64_t *MtStream::device_poly(unsigned m, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr unsigned kMaxPolys = 16;
    if (m >= kMaxPolys)
        throw std::runtime_error("mt19937_64 jump distance out of range");
    const size_t bytes = sizeof(uint64_t) * kJumpPolyWords;
    uint64_t *base = static_cast<uint64_t *>(poly_.reserve(bytes * kMaxPolys));
    while (polys_uploaded_ <= m)
    {
        const Gf2Poly &g = jump_poly(polys_uploaded_);
        std::vector<uint64_t> padded(kJumpPolyWords, 0);
        std::copy(g.begin(), g.begin() + kMtWords, padded.begin());
        check(hipMemcpyAsync(base + static_cast<size_t>(polys_uploaded_) * kJumpPolyWords, padded.data(), bytes,
                             hipMemcpyHostToDevice, s),
              "upload poly");
        check(hipStreamSynchronize(s), "sync"); // `padded` is a local buffer
        ++polys_uploaded_;
    }
    return base + static_cast<size_t>(m) * kJumpPolyWords;
}

const uint64_t *MtStream::generate(uint64_t first, uint64_t count, void *stream, int buffer)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (count == 0)
        count = 1;
    const uint64_t c_lo = first / kChunkWords;
    const uint64_t c_hi = (first + count + kChunkWords - 1) / kChunkWords;
    ensure_states(c_lo, c_hi, stream);
    const uint32_t n = static_cast<uint32_t>(c_hi - c_lo);
    uint64_t *st = static_cast<uint64_t *>(states_.get());
    uint64_t *raw = static_cast<uint64_t *>(raw_[buffer & 1].reserve(sizeof(uint64_t) * kChunkWords * n));
    const uint64_t base = table_.base();
    if (n == 1)
    {
        // a short request inside one chunk (single frames, small batches): only the prefix of the chunk that is asked
        // for is generated (whole twist rounds of 312 words); the state after the chunk is then not produced
        const uint64_t need = (first + count - c_lo * kChunkWords + kMtWords - 1) / kMtWords * kMtWords;
        if (need < kChunkWords)
        {
            check(launch_mt_generate(st + (c_lo - base) * kMtWords, nullptr, raw, 1, static_cast<uint32_t>(need), 1, s), "mt_generate");
            return raw + (first - c_lo * kChunkWords);
        }
    }
    // the state that follows the last chunk comes for free, when the table has a row for it (rows 0..kStateCap)
    const int64_t nr = table_.next_row(c_hi);
    uint64_t *next_last = nr >= 0 ? st + static_cast<size_t>(nr) * kMtWords : nullptr;
    check(launch_mt_generate(st + (c_lo - base) * kMtWords, next_last, raw, n, static_cast<uint32_t>(kChunkWords), pack_, s),
          "mt_generate");
    if (next_last)
        table_.note_next_written(c_hi);
    return raw + (first - c_lo * kChunkWords);
}

// ---------------------------------------------------------------------------------------------
Engine::Engine(const std::string &pc_file, const std::string &gen_file, int device) : device_(device)
{
    code_ = std::make_unique<LdpcCode>(pc_file, gen_file);
    if (code_->min_cn_degree() < 2)
        throw std::runtime_error("check nodes of degree < 2 are not supported (undefined in the reference decoder)");
    plan_ = build_plan(*code_);
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
        if (reg_plan_.ok)
            noise_.set_pack(4);
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
    if (code_->has_G())
    {
        std::vector<uint32_t> cp(code_->G.cptr.begin(), code_->G.cptr.end()), cr(code_->G.crow.begin(), code_->G.crow.end());
        g_col_ptr_ = static_cast<const uint32_t *>(up(cp.data(), cp.size() * 4));
        g_col_row_ = static_cast<const uint32_t *>(up(cr.data(), cr.size() * 4));
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
    if (plan_.lds_ok || reg_plan_.ok)
        return 1u << 17;
    const uint64_t per_frame = 8ull * plan_.nnz + 8ull * plan_.nc + plan_.nnz;
    return std::max<uint64_t>(1, std::min<uint64_t>(1u << 17, (8ull << 30) / per_frame));
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
        // the caller asked for the non-parity binary32 sum-product; never chosen by itself
        if (!fast_mode_supported(dev_, plan_.max_cn_degree) || plan_.has_isolated_vn)
            throw std::runtime_error("fast mode: this code is outside what the binary32 kernel takes (check nodes up to degree 8, "
                                     "nc <= 8192, LDS-resident, no isolated variable node)");
        a.ws_hb = static_cast<uint8_t *>(ws_hb_.reserve(n * nc));
        check(launch_decode_fast(a, plan_.max_cn_degree, s), "decode (fast mode, binary32)");
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
    const auto launch = [&] {
        if (plan_.lds_ok)
        {
            // Input LLRs: in registers when that frees the LDS for one more resident frame per CU (n=1024 code:
            // 40 KB -> 31 KB, five frames instead of four) and the plan allows it; in LDS otherwise.  Device memory
            // (mode 1) also reaches five frames but pays for it in memory reads (measured: no net gain).
            const size_t cu_lds = 160 * 1024, with_llr = plan_.lds_bytes, without = plan_.lds_bytes - 8 * nc;
            int llr_mode = 0;
            if (cu_lds / without > cu_lds / with_llr && plan_.vn_work_stride <= 8 && !plan_.has_isolated_vn &&
                plan_.nc <= plan_.nnz)
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
        if (handover)
            a.redo_iter_in = redo + 1 + n;
    }
    launch();
    a.redo_count_in = nullptr, a.redo_list_in = nullptr, a.redo_iter_in = nullptr, a.ws_handover = nullptr;
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
    x_ = x;
    frame_pos_ = 0;
    pair_next_ = 0;
    raw_next_ = 0;
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

uint64_t Engine::stream_raw_draws() const { return raw_next_; }

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

// Locate the accepted polar pairs that supply the normals of frames [frame_pos_, frame_pos_+n).
void Engine::awgn_prepare(uint64_t n, DecodeArgs &a, void *stream)
{
    hipStream_t user = static_cast<hipStream_t>(stream);
    ensure_rng_stream();
    hipStream_t s = static_cast<hipStream_t>(rng_stream_);
    const int buf = pp_;
    pp_ ^= 1;
    // the decode kernel that last read this pairs buffer (two batches ago) must be done before it is refilled
    if (pairs_in_use_[buf])
        check(hipStreamWaitEvent(s, static_cast<hipEvent_t>(ev_pairs_free_[buf]), 0), "wait pairs free");
    const uint64_t nct = static_cast<uint64_t>(plan_.nct);
    const uint64_t g0 = frame_pos_ * nct, g1 = g0 + n * nct; // normals [g0, g1)
    const uint64_t q_hi = (g1 - 1) >> 1;
    const uint64_t want = q_hi + 1 - pair_next_;
    // pairs[0] holds the carry pair (rank pair_next_-1), new pairs follow
    uint64_t *pairs = static_cast<uint64_t *>(pairs_[buf].reserve(16 * (want + 1)));
    uint64_t *carry = static_cast<uint64_t *>(carry_.reserve(16));
    if (pair_next_ > 0)
        check(hipMemcpyAsync(pairs, carry, 16, hipMemcpyDeviceToDevice, s), "carry in");
    ScanResult *res = static_cast<ScanResult *>(scan_result_.reserve(sizeof(ScanResult)));
    uint64_t trials = static_cast<uint64_t>(want * 1.2732395447351628 + 8.0 * std::sqrt(static_cast<double>(want)) + 256);
    ScanResult h{};
    PhaseTrace tr;
    prof_mark(1, s);
    for (;;)
    {
        const uint64_t *raw = noise_.generate(raw_next_, 2 * trials, s);
        tr.mark("generate(enqueue)");
        const uint32_t n_blocks = static_cast<uint32_t>((trials + kScanBlock - 1) / kScanBlock);
        uint32_t *counts = static_cast<uint32_t *>(scan_counts_.reserve(4 * static_cast<size_t>(n_blocks)));
        uint64_t *offs = static_cast<uint64_t *>(scan_offsets_.reserve(8 * static_cast<size_t>(n_blocks)));
        check(launch_polar_scan(raw, trials, want, counts, offs, pairs + 2, res, s), "polar_scan");
        check(hipMemcpyAsync(&h, res, sizeof h, hipMemcpyDeviceToHost, s), "scan result");
        check(hipStreamSynchronize(s), "sync"); // waits for the noise-stream kernels only, not for the caller's decode
        tr.mark("scan+sync");
        if (h.enough)
            break;
        trials += trials / 8 + 4096; // vanishingly rare: take a longer look at the same stream
    }
    prof_mark(1, s);
    check(hipMemcpyAsync(carry, pairs + 2 * want, 16, hipMemcpyDeviceToDevice, s), "carry out");
    check(hipEventRecord(static_cast<hipEvent_t>(ev_pairs_ready_[buf]), s), "event");
    check(hipStreamWaitEvent(user, static_cast<hipEvent_t>(ev_pairs_ready_[buf]), 0), "wait pairs ready");
    a.pairs = pairs;
    a.pair_base = pair_next_ - 1; // wraps to 2^64-1 for the very first batch: q - pair_base == q + 1
    a.normal_base = g0;
    a.sigma = sigma_, a.sigma2 = sigma2_;
    a.shorten_llr = 99999.9; // channel.cpp:83
    a.pairs_buffer = buf;
    pair_next_ += want;
    raw_next_ += 2 * h.trials_used;
}

void Engine::stream_skip(uint64_t n_frames, void *stream)
{
    if (!chan_)
        throw std::runtime_error("stream_begin() has not been called");
    upload_plan();
    const uint64_t nct = static_cast<uint64_t>(plan_.nct);
    while (n_frames)
    {
        const uint64_t n = std::min<uint64_t>(n_frames, 1u << 17);
        encode_frames(n, false, stream);
        if (chan_ == kAwgn)
        {
            DecodeArgs a{};
            awgn_prepare(n, a, stream);
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
uint64_t Engine::shard_capacity(uint64_t target_frames, int world)
{
    const uint64_t per = (target_frames + world - 1) / world;
    return per + per / 16 + 64; // the acceptance count of a piece varies by a few parts in a thousand
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
                                                void *stream)
{
    if (!chan_)
        throw std::runtime_error("stream_begin() has not been called");
    upload_plan();
    const int R = comm.world(), r = comm.rank();
    const uint64_t nct = static_cast<uint64_t>(plan_.nct), cap = shard_capacity(target_frames, R);
    if (cap > max_sub_batch())
        throw std::runtime_error("sharded step too large for one launch per rank");
    target_frames = std::max<uint64_t>(target_frames, 1);
    if (stream_mode_ == 1)
        throw std::runtime_error("stream_decode_sharded after stream_decode on the same stream: call stream_begin first");
    stream_mode_ = 2;
    ShardStep st;
    st.step_first = frame_pos_;
    DecodeArgs a{};
    if (chan_ == kAwgn)
    {
        ensure_rng_stream();
        hipStream_t s = static_cast<hipStream_t>(rng_stream_), user = static_cast<hipStream_t>(stream);
        const int buf = pp_;
        pp_ ^= 1;
        if (pairs_in_use_[buf])
            check(hipStreamWaitEvent(s, static_cast<hipEvent_t>(ev_pairs_free_[buf]), 0), "wait pairs free");
        // the step: T trials of the raw stream, a multiple of R scan blocks, enough for about target_frames frames
        const uint64_t want_pairs = (target_frames * nct + 1) / 2;
        const uint64_t unit = static_cast<uint64_t>(R) * kScanBlock;
        const uint64_t T = std::max<uint64_t>(unit, static_cast<uint64_t>(static_cast<double>(want_pairs) * 1.2732395447351628) / unit * unit);
        const uint64_t piece = T / R;
        // margin: the pairs of one frame beyond the piece (the last frame a rank owns may end in its neighbour's piece)
        const uint64_t margin = ((nct / 2 + 2) * 2 + 4 * kScanBlock - 1) / kScanBlock * kScanBlock;
        const uint64_t t0 = raw_next_ / 2 + static_cast<uint64_t>(r) * piece;
        prof_mark(1, s);
        const uint64_t *raw = noise_.generate(2 * t0, 2 * (piece + margin), s);
        const uint32_t n_blocks = static_cast<uint32_t>((piece + margin) / kScanBlock);
        uint32_t *counts = static_cast<uint32_t *>(scan_counts_.reserve(4 * static_cast<size_t>(n_blocks)));
        uint64_t *offs = static_cast<uint64_t *>(scan_offsets_.reserve(8 * static_cast<size_t>(n_blocks)));
        ScanResult *res = static_cast<ScanResult *>(scan_result_.reserve(2 * sizeof(ScanResult)));
        check(launch_polar_count(raw, piece + margin, piece, counts, offs, res, res + 1, s), "polar_count");
        ScanResult h[2];
        check(hipMemcpyAsync(h, res, sizeof h, hipMemcpyDeviceToHost, s), "scan result");
        check(hipStreamSynchronize(s), "sync");
        // the exchange: every rank's accepted-pair count of its piece -> where each piece starts in the pair sequence
        std::vector<uint64_t> acc(R);
        const uint64_t mine = h[0].accepted;
        comm.all_gather(&mine, acc.data(), sizeof mine);
        std::vector<uint64_t> P(R + 1);
        P[0] = pair_next_;
        for (int q = 0; q < R; ++q)
            P[q + 1] = P[q] + acc[q];
        auto first_frame = [&](uint64_t pair) { return (2 * pair + nct - 1) / nct; }; // first frame whose first pair is >= pair
        st.first = first_frame(P[r]);
        st.n = first_frame(P[r + 1]) - st.first;
        st.step_frames = first_frame(P[R]) - st.step_first;
        if (first_frame(P[0]) != frame_pos_)
            throw std::runtime_error("sharded stream out of step");
        if (st.n > cap)
            throw std::runtime_error("sharded step: more frames in a piece than the output buffers hold");
        if (st.n)
        {
            const uint64_t last_pair = ((st.first + st.n) * nct - 1) >> 1;
            const uint64_t need = last_pair - P[r] + 1;
            if (need > h[1].accepted)
                throw std::runtime_error("sharded step: a frame extends beyond the margin scanned after the piece");
            uint64_t *pairs = static_cast<uint64_t *>(pairs_[buf].reserve(16 * (need + 1)));
            check(launch_polar_compact(raw, piece + margin, offs, need, pairs, res, s), "polar_compact");
            a.pairs = pairs;
            a.pair_base = P[r];
        }
        prof_mark(1, s);
        check(hipEventRecord(static_cast<hipEvent_t>(ev_pairs_ready_[buf]), s), "event");
        check(hipStreamWaitEvent(user, static_cast<hipEvent_t>(ev_pairs_ready_[buf]), 0), "wait pairs ready");
        a.mode = kModeAwgn;
        a.normal_base = st.first * nct;
        a.sigma = sigma_, a.sigma2 = sigma2_;
        a.shorten_llr = 99999.9; // channel.cpp:83
        a.pairs_buffer = buf;
        pair_next_ = P[R];
        raw_next_ += 2 * T;
    }
    else
    {
        st.step_frames = target_frames;
        const uint64_t base = target_frames / R, extra = target_frames % R;
        st.n = base + (static_cast<uint64_t>(r) < extra ? 1 : 0);
        st.first = st.step_first + base * r + std::min<uint64_t>(r, extra);
    }
    // encoder: every rank walks the whole step's info words (kc draws per frame, a tenth of the noise stream's) so
    // that all of them hold the same accumulated codeword afterwards; codewords are formed for the own frames only
    const uint8_t *cw = nullptr;
    if (code_->has_G())
    {
        for (uint64_t left = st.first - st.step_first; left;)
        {
            const uint64_t n = std::min<uint64_t>(left, 1u << 17);
            encode_frames(n, false, stream);
            left -= n;
        }
        cw = encode_frames(st.n, true, stream);
        for (uint64_t left = st.step_first + st.step_frames - (st.first + st.n); left;)
        {
            const uint64_t n = std::min<uint64_t>(left, 1u << 17);
            encode_frames(n, false, stream);
            left -= n;
        }
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
