// api.cpp — the C ABI of libldpc.so (include/ldpc_amd.h).
//
// Part 1 mirrors the reference's src/shared.cpp:9-78 (same symbols, same struct layouts, same
// process-global state and error behaviour: failures print a message and exit(EXIT_FAILURE)).
// Part 2 is the batch interface over ldpc_amd::Engine.
#include "../../include/ldpc_amd.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "engine.hpp"
#include "shard_place.hpp"
#include "sim.hpp"

using namespace ldpc_amd;

struct ldpc_hip_comm
{
    std::unique_ptr<Comm> comm;
    std::string description;
};

struct ldpc_hip_ctx
{
    std::unique_ptr<Engine> eng;
    MtStream aux;
    DeviceBuffer aux_out;
    std::string description;
};

namespace
{
thread_local std::string g_err;

std::unique_ptr<Engine> g_engine; // shared.cpp:4-5: one code + one decoder per process
bool g_sticky_min_sum = false;    // decoder.h:73-80: set_param never switches back from BP_MS

DecParams to_params(const decoder_param &p)
{
    DecParams d;
    d.early_term = p.earlyTerm;
    d.iterations = p.iterations;
    d.min_sum = p.type && std::strcmp(p.type, "BP_MS") == 0;
    return d;
}

int channel_from(const char *type)
{
    if (type && !std::strcmp(type, "AWGN"))
        return kAwgn;
    if (type && !std::strcmp(type, "BSC"))
        return kBsc;
    if (type && !std::strcmp(type, "BEC"))
        return kBec;
    return 0;
}

[[noreturn]] void die(const char *where, const std::exception &e)
{
    std::cout << "Error: " << where << " " << e.what() << std::endl;
    std::exit(EXIT_FAILURE);
}

template <typename F>
int guarded(F &&f)
{
    try
    {
        f();
        return 0;
    }
    catch (const std::exception &e)
    {
        g_err = e.what();
        return -1;
    }
}

BatchOut to_out(const ldpc_hip_out *o)
{
    BatchOut b;
    if (o)
    {
        b.iters = o->iters, b.bit_errors = o->bit_errors, b.hard = o->hard;
        b.llr_out = o->llr_out, b.llr_in = o->llr_in, b.codeword = o->codeword;
    }
    return b;
}
} // namespace

extern "C"
{

// ------------------------------------------------------------------------------------------------
// Part 1 — reference ABI
// ------------------------------------------------------------------------------------------------
void ldpc_setup(const char *pcFile, const char *genFile, int *n, int *m, int *nct, int *mct)
{
    try
    {
        // the six reference symbols carry no device argument: LDPC_AMD_DEVICE picks the GPU of the process-global engine
        // (read at every ldpc_setup; default 0), e.g. one pyLDPC process per GPU of an 8-GPU node
        int device = 0;
        if (const char *e = std::getenv("LDPC_AMD_DEVICE"))
        {
            char *end = nullptr;
            const long v = std::strtol(e, &end, 10);
            if (end == e || *end != '\0' || v < 0 || v > 1023)
                throw std::runtime_error(std::string("LDPC_AMD_DEVICE is not a device index: ") + e);
            device = static_cast<int>(v);
        }
        g_engine = std::make_unique<Engine>(pcFile ? pcFile : "", genFile ? genFile : "", device);
    }
    catch (const std::exception &e)
    {
        die("ldpc_code():", e); // ldpc.cpp:16-20
    }
    g_sticky_min_sum = false;
    const LdpcCode &c = g_engine->code();
    *n = c.nc(), *m = c.mc(), *nct = c.nct(), *mct = c.mct();
}

void simulate(decoder_param decoderParams, channel_param channelParam, simulation_param simParam,
              sim_results_t *results, bool *stopFlag)
{
    SimRequest rq;
    rq.dec = to_params(decoderParams);
    rq.channel = channel_from(channelParam.type);
    rq.seed = channelParam.seed;
    for (int i = 0; i < 3; ++i)
        rq.x_range[i] = channelParam.xRange[i];
    rq.max_frames = simParam.maxFrames;
    rq.min_fec = simParam.fec;
    rq.result_file = simParam.resultFile ? simParam.resultFile : "";
    rq.cli_output = false; // the reference library is built with LIB_SHARED (CMakeLists.txt:22)
    try
    {
        if (!g_engine)
            throw std::runtime_error("ldpc_setup() has not been called");
        if (!rq.channel)
            throw std::runtime_error("No channel selected."); // ldpcsim.cpp:74
        g_engine->bec_deg1_compat = std::getenv("LDPC_AMD_BEC_COMPAT") != nullptr;
        run_simulation(*g_engine, rq, results, nullptr, stopFlag);
    }
    catch (const std::exception &e)
    {
        die("ldpc_sim::ldpc_sim()", e);
    }
}

int calculate_rank(void) { return g_engine ? g_engine->code().H.rank() : 0; }

void encode(uint8_t *infoWord, uint8_t *codeWord)
{
    if (!g_engine)
        return;
    const LdpcCode &c = g_engine->code();
    std::vector<uint8_t> cw(std::max(c.nc(), c.G.cols), 0);
    if (c.has_G())
    {
        // the reference reads kct() info bits and multiplies by G (kc() rows)
        std::vector<uint8_t> u(std::max(c.G.rows, c.kct()), 0);
        for (int i = 0; i < c.kct(); ++i)
            u[i] = infoWord[i] != 0;
        c.G.multiply_left(u.data(), cw.data());
    }
    else
        std::cout << "Error: encode() no generator matrix loaded" << std::endl;
    for (int i = 0; i < c.nct(); ++i)
        codeWord[i] = cw[c.bit_pos[i]];
}

int decode(decoder_param decoderParams, double *llr, double *llrOut)
{
    try
    {
        if (!g_engine)
            throw std::runtime_error("ldpc_setup() has not been called");
        const LdpcCode &c = g_engine->code();
        DecParams p = to_params(decoderParams);
        g_sticky_min_sum = g_sticky_min_sum || p.min_sum;
        p.min_sum = g_sticky_min_sum;
        std::vector<double> in(c.nc(), 0.0), out(c.nc(), 0.0); // punctured and shortened stay 0.0 (shared.cpp:50)
        for (int i = 0; i < c.nct(); ++i)
            in[c.bit_pos[i]] = llr[i];
        uint32_t iters = 0;
        BatchOut o;
        o.iters = &iters;
        o.llr_out = out.data();
        g_engine->decode_llr(p, 1, in.data(), o, nullptr);
        for (int i = 0; i < c.nct(); ++i)
            llrOut[i] = out[c.bit_pos[i]];
        return static_cast<int>(iters);
    }
    catch (const std::exception &e)
    {
        die("decode()", e);
    }
}

void syndrome(uint8_t *word, uint8_t *synd)
{
    if (!g_engine)
        return;
    const LdpcCode &c = g_engine->code();
    std::vector<uint8_t> s(c.mc(), 0);
    c.H.multiply_right(word, s.data());
    std::memcpy(synd, s.data(), s.size());
}

// ------------------------------------------------------------------------------------------------
// Part 2 — batch interface
// ------------------------------------------------------------------------------------------------
int ldpc_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
    {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *ldpc_hip_last_error(void) { return g_err.c_str(); }

ldpc_hip_ctx *ldpc_hip_create(const char *pcFile, const char *genFile, int device)
{
    ldpc_hip_ctx *ctx = nullptr;
    if (guarded([&] {
            auto c = std::make_unique<ldpc_hip_ctx>();
            c->eng = std::make_unique<Engine>(pcFile ? pcFile : "", genFile ? genFile : "", device);
            ctx = c.release();
        }) != 0)
        return nullptr;
    return ctx;
}

void ldpc_hip_destroy(ldpc_hip_ctx *ctx) { delete ctx; }

void ldpc_hip_code_info(const ldpc_hip_ctx *ctx, int64_t info[10])
{
    const LdpcCode &c = ctx->eng->code();
    const Plan &p = ctx->eng->plan();
    info[0] = c.nc(), info[1] = c.mc(), info[2] = c.nnz(), info[3] = c.nct(), info[4] = c.mct();
    info[5] = c.kct(), info[6] = c.kc(), info[7] = c.max_degree, info[8] = p.lds_ok;
    info[9] = static_cast<int64_t>(p.lds_bytes);
    if (!p.lds_ok && ctx->eng->reg2_plan().ok)
        info[8] = 3; // register-resident decoder, totals form (kernels_reg2.hip)
    else if (!p.lds_ok && ctx->eng->reg_plan().ok)
        info[8] = 2; // register-resident decoder, messages form (kernels_reg.hip)
}

const char *ldpc_hip_describe(ldpc_hip_ctx *ctx)
{
    ctx->description = ctx->eng->code().describe();
    return ctx->description.c_str();
}

void ldpc_hip_set_bec_compat(ldpc_hip_ctx *ctx, int compat) { ctx->eng->bec_deg1_compat = compat != 0; }
void ldpc_hip_set_fast_mode(ldpc_hip_ctx *ctx, int mode) { ctx->eng->fast_mode = mode < 0 || mode > 3 ? 0 : mode; }

int ldpc_hip_decode_batch(ldpc_hip_ctx *ctx, decoder_param dec, uint64_t n, const double *llr_in,
                          const ldpc_hip_out *out, void *hip_stream)
{
    return guarded([&] { ctx->eng->decode_llr(to_params(dec), n, llr_in, to_out(out), hip_stream); });
}

int ldpc_hip_stream_begin(ldpc_hip_ctx *ctx, int channel, uint64_t seed, double x)
{
    return guarded([&] { ctx->eng->stream_begin(channel, seed, x, true); });
}

int ldpc_hip_stream_skip(ldpc_hip_ctx *ctx, uint64_t n, void *hip_stream)
{
    return guarded([&] { ctx->eng->stream_skip(n, hip_stream); });
}

int ldpc_hip_stream_decode(ldpc_hip_ctx *ctx, decoder_param dec, uint64_t n, const ldpc_hip_out *out,
                           void *hip_stream)
{
    return guarded([&] { ctx->eng->stream_decode(to_params(dec), n, to_out(out), hip_stream); });
}

uint64_t ldpc_hip_stream_frame(const ldpc_hip_ctx *ctx) { return ctx->eng->stream_frame(); }
uint64_t ldpc_hip_stream_raw_draws(const ldpc_hip_ctx *ctx)
{
    uint64_t n = ~0ull; // (AWGN: one generator pass locates the last consumed trial; UINT64_MAX on error)
    guarded([&] { n = ctx->eng->stream_raw_draws(); });
    return n;
}

int ldpc_hip_synchronize(ldpc_hip_ctx *ctx, void *hip_stream)
{
    return guarded([&] { ctx->eng->synchronize(hip_stream); });
}

void ldpc_hip_set_profiling(ldpc_hip_ctx *ctx, int on)
{
    guarded([&] { ctx->eng->set_profiling(on != 0); });
}

float ldpc_hip_last_ms(ldpc_hip_ctx *ctx, int which) { return ctx->eng->last_ms(which); }

int ldpc_hip_mt64(ldpc_hip_ctx *ctx, uint64_t seed, uint64_t first, uint64_t n, uint64_t *out, void *hip_stream)
{
    return guarded([&] {
        if (ldpc_hip_device_count() <= ctx->eng->device())
            throw std::runtime_error("no usable HIP device (MI355X required)");
        if (hipSetDevice(ctx->eng->device()) != hipSuccess)
            throw std::runtime_error("hipSetDevice failed");
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        ctx->aux.reset(seed);
        uint64_t done = 0;
        while (done < n)
        {
            const uint64_t k = std::min<uint64_t>(n - done, 64ull << 20);
            const uint64_t *raw = ctx->aux.generate(first + done, k, hip_stream);
            hipPointerAttribute_t attr;
            bool dev = hipPointerGetAttributes(&attr, out) == hipSuccess && attr.type == hipMemoryTypeDevice;
            if (!dev)
                (void)hipGetLastError();
            if (hipMemcpyAsync(out + done, raw, 8 * k, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s) !=
                    hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                throw std::runtime_error("copy of mt19937_64 words failed");
            done += k;
        }
    });
}

int ldpc_hip_batch_counters(ldpc_hip_ctx *ctx, const uint32_t *iters, const uint32_t *bit_errors, uint64_t n,
                            uint32_t max_iters, int early_term, int64_t *counters, void *hip_stream)
{
    return guarded([&] {
        if (ldpc_hip_device_count() <= ctx->eng->device())
            throw std::runtime_error("no usable HIP device (MI355X required)");
        if (hipSetDevice(ctx->eng->device()) != hipSuccess)
            throw std::runtime_error("hipSetDevice failed");
        if (!iters || !bit_errors || !counters)
            throw std::runtime_error("ldpc_hip_batch_counters: null pointer");
        if (ldpc_amd::launch_batch_counters(iters, bit_errors, n, max_iters, early_term,
                                            reinterpret_cast<long long *>(counters), hip_stream) != hipSuccess)
            throw std::runtime_error("ldpc_hip_batch_counters: launch failed");
    });
}

int ldpc_hip_selftest_division(ldpc_hip_ctx *ctx, uint64_t n, uint64_t seed, uint64_t *mismatches)
{
    return guarded([&] {
        if (ldpc_hip_device_count() <= ctx->eng->device())
            throw std::runtime_error("no usable HIP device (MI355X required)");
        if (hipSetDevice(ctx->eng->device()) != hipSuccess)
            throw std::runtime_error("hipSetDevice failed");
        unsigned long long *d = nullptr, h = 0;
        if (hipMalloc(&d, sizeof h) != hipSuccess || hipMemset(d, 0, sizeof h) != hipSuccess)
            throw std::runtime_error("hipMalloc failed");
        const int rc = ldpc_amd::launch_division_selftest(n, seed, d, nullptr);
        const bool ok = rc == hipSuccess && hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d);
        if (!ok)
            throw std::runtime_error("division self-test failed to run");
        *mismatches = h;
    });
}

int ldpc_hip_selftest_math(ldpc_hip_ctx *ctx, int fn, uint64_t n, const double *a, const double *b, double *out)
{
    return guarded([&] {
        if (ldpc_hip_device_count() <= ctx->eng->device())
            throw std::runtime_error("no usable HIP device (MI355X required)");
        if (hipSetDevice(ctx->eng->device()) != hipSuccess)
            throw std::runtime_error("hipSetDevice failed");
        const int w = ldpc_amd::math_selftest_width(fn);
        if (w == 0 || !a || !out)
            throw std::runtime_error("ldpc_hip_selftest_math: unknown function or null buffer");
        const size_t bytes = 8 * n * static_cast<size_t>(w);
        double *da = nullptr, *db = nullptr, *dout = nullptr;
        bool ok = hipMalloc(&da, bytes) == hipSuccess && hipMalloc(&dout, bytes) == hipSuccess &&
                  hipMemcpy(da, a, bytes, hipMemcpyHostToDevice) == hipSuccess;
        if (ok && b)
            ok = hipMalloc(&db, bytes) == hipSuccess && hipMemcpy(db, b, bytes, hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && ldpc_amd::launch_math_selftest(fn, n, da, db ? db : da, dout, nullptr) == hipSuccess &&
             hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(da), (void)hipFree(db), (void)hipFree(dout);
        if (!ok)
            throw std::runtime_error("ldpc_hip_selftest_math failed to run");
    });
}

// host-only replay of the chunk-state bookkeeping (mtstates.hpp) on a symbolic table: every row records which chunk's state
// it holds; an operation that reads a row without a valid state, or a request that ends without its rows, fails the replay
namespace
{
struct SymbolicTable
{
    std::vector<int64_t> row; // chunk id, -1 = nothing valid
    uint64_t tasks = 0, launches = 0;
    explicit SymbolicTable(size_t n) : row(n, -1) {}
    bool apply(const std::vector<StateOp> &ops)
    {
        for (const StateOp &op : ops)
            switch (op.kind)
            {
            case StateOp::kUpload0: row[op.dst] = 0; break;
            case StateOp::kCopy:
                if (row[op.src] < 0)
                    return false;
                row[op.dst] = row[op.src];
                break;
            case StateOp::kJump:
            {
                std::vector<int64_t> src(op.n);
                for (uint32_t i = 0; i < op.n; ++i) // all tasks of a launch may read before any of them writes, or after
                {
                    src[i] = row[(op.src + i) % op.mod];
                    if (src[i] < 0)
                        return false;
                    for (uint32_t k = 0; k < op.n; ++k) // a task's source must not be another task's destination
                        if (k != i && (op.dst + k) % op.mod == (op.src + i) % op.mod)
                            return false;
                }
                for (uint32_t i = 0; i < op.n; ++i)
                    row[(op.dst + i) % op.mod] = src[i] + static_cast<int64_t>(op.stride);
                tasks += op.n, ++launches;
                break;
            }
            }
        return true;
    }
};
} // namespace

uint64_t ldpc_hip_selftest_chunk_table(uint64_t first_chunk, uint64_t chunks_per_request, uint64_t n_requests, uint64_t gap)
{
    StateRing t;
    SymbolicTable sym(StateRing::kTotalRows);
    uint64_t c = first_chunk;
    std::vector<StateOp> ops;
    try
    {
        for (uint64_t i = 0; i < n_requests; ++i, c += chunks_per_request + gap)
        {
            ops.clear();
            t.ensure(c, c + chunks_per_request, ops);
            if (!sym.apply(ops))
                return ~0ull;
            for (uint64_t k = c; k < c + chunks_per_request; ++k)
                if (sym.row[k % StateRing::kRows] != static_cast<int64_t>(k))
                    return ~0ull; // a requested chunk has no valid row
            if (gap == 0 && i % 3 != 2) // the engine's look-ahead for a reader going front to back: two requests ahead
            {
                ops.clear();
                t.extend_to(c + 3 * chunks_per_request + 2, ops);
                if (!sym.apply(ops))
                    return ~0ull;
                for (uint64_t k = c; k < c + chunks_per_request; ++k) // (never at the expense of the request's own rows)
                    if (sym.row[k % StateRing::kRows] != static_cast<int64_t>(k))
                        return ~0ull;
            }
        }
    }
    catch (const std::exception &e)
    {
        g_err = e.what();
        return ~0ull;
    }
    return sym.tasks;
}

uint64_t ldpc_hip_selftest_shard_table(int world, int rank, uint32_t piece_chunks, uint64_t steps, uint64_t *launches)
{
    StridedTable t;
    SymbolicTable sym(StridedTable::kTotalRows);
    std::vector<StateOp> ops;
    const uint32_t n = piece_chunks + 1;
    const uint64_t stride = static_cast<uint64_t>(world) * piece_chunks;
    uint64_t tasks_after_first = 0, launches_after_first = 0;
    try
    {
        for (uint64_t s = 0; s < steps; ++s)
        {
            const uint64_t first = s * stride + static_cast<uint64_t>(rank) * piece_chunks;
            ops.clear();
            const uint32_t base = t.position(first, n, stride, ops);
            const uint64_t t0 = sym.tasks, l0 = sym.launches;
            if (!sym.apply(ops))
                return ~0ull;
            for (uint32_t i = 0; i < n; ++i)
                if (sym.row[base + i] != static_cast<int64_t>(first + i))
                    return ~0ull;
            // the look-ahead of the engine (the tables of the next two steps) after two steps in three: a step finds its table
            // ready (no operation of its own) or pays its one launch itself
            if (s % 3 != 2)
            {
                ops.clear();
                t.look_ahead(ops);
                if (!sym.apply(ops))
                    return ~0ull;
                for (uint32_t i = 0; i < n; ++i) // (the table in use is untouched)
                    if (sym.row[base + i] != static_cast<int64_t>(first + i))
                        return ~0ull;
            }
            if (s > 0)
                tasks_after_first += sym.tasks - t0, launches_after_first += sym.launches - l0;
        }
    }
    catch (const std::exception &e)
    {
        g_err = e.what();
        return ~0ull;
    }
    if (launches)
        *launches = launches_after_first;
    return tasks_after_first;
}

int ldpc_hip_comm_unique_id(uint8_t id[128])
{
    return guarded([&] { rccl_unique_id(id); });
}

ldpc_hip_comm *ldpc_hip_comm_create(int rank, int world, int device, const uint8_t id[128])
{
    ldpc_hip_comm *c = nullptr;
    if (guarded([&] {
            auto p = std::make_unique<ldpc_hip_comm>();
            p->comm = make_rccl_comm(rank, world, device, id);
            c = p.release();
        }) != 0)
        return nullptr;
    return c;
}

ldpc_hip_comm *ldpc_hip_comm_create_shm(int rank, int world, const char *name)
{
    ldpc_hip_comm *c = nullptr;
    if (guarded([&] {
            auto p = std::make_unique<ldpc_hip_comm>();
            p->comm = make_shm_comm(rank, world, name ? name : "/ldpc_amd");
            c = p.release();
        }) != 0)
        return nullptr;
    return c;
}

ldpc_hip_comm *ldpc_hip_comm_create_echo(int rank, int world)
{
    ldpc_hip_comm *c = nullptr;
    if (guarded([&] {
            auto p = std::make_unique<ldpc_hip_comm>();
            p->comm = make_echo_comm(rank, world);
            c = p.release();
        }) != 0)
        return nullptr;
    return c;
}

void ldpc_hip_comm_destroy(ldpc_hip_comm *comm) { delete comm; }

int ldpc_hip_comm_allgather(ldpc_hip_comm *comm, const void *send, void *recv, uint64_t bytes)
{
    return guarded([&] { comm->comm->all_gather(send, recv, bytes); });
}

void ldpc_hip_fused_plan_info(const ldpc_hip_ctx *ctx, int64_t info[8])
{
    const FusedPlan &f = ctx->eng->fused_plan();
    info[0] = f.ok ? 1 : 0, info[1] = f.n_slots, info[2] = f.vnb, info[3] = f.cnl, info[4] = f.wide_exclusive ? 1 : 0;
    info[5] = f.calls_stride, info[6] = f.has_shortened ? 1 : 0, info[7] = static_cast<int64_t>(f.vn_slot.size());
}

int ldpc_hip_selftest_layer_plan(ldpc_hip_ctx *ctx, int32_t *step_of_row)
{
    int n = -1;
    if (guarded([&] {
            const LayerPlan L = build_layer_plan(ctx->eng->code(), ctx->eng->plan());
            if (!L.ok)
                throw std::runtime_error("the layered schedule does not take this code");
            for (size_t i = 0; i < L.step_of_row.size(); ++i)
                step_of_row[i] = L.step_of_row[i];
            n = static_cast<int>(L.steps.size());
        }) != 0)
        return -1;
    return n;
}

int ldpc_hip_selftest_place(ldpc_hip_comm *comm, uint64_t nct, uint64_t pairs_before, uint64_t frame_pos, uint64_t cap,
                            uint64_t piece_pairs, uint64_t pairs_with_margin, uint64_t status, uint64_t out[5])
{
    return guarded([&] {
        const int R = comm->comm->world(), r = comm->comm->rank();
        const uint64_t send[3] = {piece_pairs, pairs_with_margin, status};
        std::vector<uint64_t> all(3 * static_cast<size_t>(R));
        comm->comm->all_gather(send, all.data(), sizeof send);
        const ShardPlacement pl = place_pieces(all.data(), R, r, pairs_before, frame_pos, nct, cap, status ? "status word set by the caller" : "");
        out[0] = pl.first, out[1] = pl.n, out[2] = pl.step_frames, out[3] = pl.pair_start, out[4] = pl.pairs_after;
    });
}

void ldpc_hip_comm_stats(ldpc_hip_comm *comm, double out[4], int reset) { comm->comm->exchange_stats(out, reset != 0); }

const char *ldpc_hip_comm_describe(ldpc_hip_comm *comm)
{
    comm->description = comm->comm->describe();
    return comm->description.c_str();
}

uint64_t ldpc_hip_shard_capacity(const ldpc_hip_ctx *ctx, uint64_t target_frames, int world) { return ctx->eng->shard_capacity(target_frames, world < 1 ? 1 : world); }
uint64_t ldpc_hip_jump_tasks(const ldpc_hip_ctx *ctx) { return ctx->eng->noise_jump_tasks(); }

int ldpc_hip_stream_decode_sharded(ldpc_hip_ctx *ctx, ldpc_hip_comm *comm, decoder_param dec, uint64_t target_frames,
                                   const ldpc_hip_out *out, uint64_t step[4], void *hip_stream)
{
    return guarded([&] {
        const Engine::ShardStep st = ctx->eng->stream_decode_sharded(*comm->comm, to_params(dec), target_frames, to_out(out), hip_stream);
        if (step)
            step[0] = st.step_first, step[1] = st.step_frames, step[2] = st.first, step[3] = st.n;
    });
}

int ldpc_hip_simulate_sharded(ldpc_hip_ctx *ctx, ldpc_hip_comm *comm, decoder_param dec, channel_param ch,
                              simulation_param sim, sim_results_t *results, uint64_t *totals, bool *stopFlag, int cli_output)
{
    int n = -1;
    int rc = guarded([&] {
        SimRequest rq;
        rq.dec = to_params(dec);
        rq.channel = channel_from(ch.type);
        if (!rq.channel)
            throw std::runtime_error("No channel selected.");
        rq.seed = ch.seed;
        for (int i = 0; i < 3; ++i)
            rq.x_range[i] = ch.xRange[i];
        rq.max_frames = sim.maxFrames;
        rq.min_fec = sim.fec;
        rq.result_file = sim.resultFile ? sim.resultFile : "";
        rq.cli_output = cli_output != 0;
        n = run_simulation(*ctx->eng, rq, results, totals, stopFlag, comm ? comm->comm.get() : nullptr);
    });
    return rc == 0 ? n : -1;
}

int ldpc_hip_simulate(ldpc_hip_ctx *ctx, decoder_param dec, channel_param ch, simulation_param sim,
                      sim_results_t *results, uint64_t *totals, bool *stopFlag, int cli_output)
{
    int n = -1;
    int rc = guarded([&] {
        SimRequest rq;
        rq.dec = to_params(dec);
        rq.channel = channel_from(ch.type);
        if (!rq.channel)
            throw std::runtime_error("No channel selected.");
        rq.seed = ch.seed;
        for (int i = 0; i < 3; ++i)
            rq.x_range[i] = ch.xRange[i];
        rq.max_frames = sim.maxFrames;
        rq.min_fec = sim.fec;
        rq.result_file = sim.resultFile ? sim.resultFile : "";
        rq.cli_output = cli_output != 0;
        n = run_simulation(*ctx->eng, rq, results, totals, stopFlag);
    });
    return rc == 0 ? n : -1;
}

} // extern "C"
