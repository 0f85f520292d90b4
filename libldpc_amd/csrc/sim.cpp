// sim.cpp — the Monte-Carlo loop of the reference (src/sim/ldpcsim.cpp:97-263) on top of the batched
// GPU step.  The reference decodes one frame per loop trip and re-evaluates its stop rule after each;
// here a batch of frames of the same noise stream is decoded per launch and the per-frame results are
// then folded in stream order with the reference's rule, so the counters (frames, fec, bec, iters) and
// every line written are those of a single-threaded reference run with the same seed.  Frames decoded
// past the stopping frame are discarded.
#include "sim.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace ldpc_amd
{

namespace
{
// ldpcsim.cpp:175-255 over `count` consecutive frames, starting from `frames0` counted frames and `fec0` frame errors
struct Fold
{
    uint64_t n = 0;            // frames walked (== count unless the stop rule fired)
    uint64_t fec = 0, bec = 0, iters = 0;
    uint64_t n_at_err = 0;     // frames walked up to and including the last frame error (0: none)
    uint64_t iters_at_err = 0; // iterations summed up to and including that frame
    uint64_t stopped = 0;
};

Fold fold_range(const uint32_t *it, const uint32_t *be, uint64_t count, uint64_t frames0, uint64_t fec0, uint64_t min_fec,
                uint64_t max_frames)
{
    Fold f;
    for (uint64_t i = 0; i < count; ++i)
    {
        f.n = i + 1;
        f.iters += it[i];
        if (fec0 + f.fec < min_fec && be[i] > 0)
        {
            f.bec += be[i];
            ++f.fec;
            f.n_at_err = i + 1, f.iters_at_err = f.iters;
        }
        if (!(fec0 + f.fec < min_fec && frames0 + f.n < max_frames)) // ldpcsim.cpp:255
        {
            f.stopped = 1;
            break;
        }
    }
    return f;
}

int run_simulation_sharded(Engine &eng, const SimRequest &rq, sim_results_t *results, uint64_t *totals, bool *stop_flag, Comm &comm)
{
    using clock = std::chrono::high_resolution_clock;
    const int R = comm.world(), me = comm.rank();
    const bool root = me == 0;
    std::vector<double> xs;
    for (double v = rq.x_range[0]; v < rq.x_range[1]; v += rq.x_range[2])
        xs.push_back(v);
    const bool eps_axis = rq.channel == kBsc || rq.channel == kBec;
    if (eps_axis)
        std::reverse(xs.begin(), xs.end());
    std::vector<std::string> lines(xs.size() + 1);
    if (rq.cli_output)
        lines[0] = "snr fer ber frames avg_iter frame_time";
    if (root)
    {
        std::cout << "=============================" << "===========================================================" << std::endl;
        std::cout << "  FEC   |      FRAME     |   " << (eps_axis ? "EPS" : "SNR")
                  << "   |    BER     |    FER     | AVGITERS  |  TIME/FRAME   \n";
        std::cout << "========+================+===" << "======+============+============+===========+==============" << std::endl;
    }
    const uint64_t nc = static_cast<uint64_t>(eng.code().nc());
    const uint64_t kNoLimit = ~0ull;
    std::vector<uint32_t> it_buf, be_buf;
    for (size_t i = 0; i < xs.size(); ++i)
    {
        uint64_t bec = 0, fec = 0, frames = 0, iters = 0;
        const auto t_start = clock::now();
        eng.stream_begin(rq.channel, rq.seed, xs[i], /*fresh=*/i == 0);
        const uint64_t msb = eng.max_sub_batch() * 3 / 4; // (a piece may hold a few per cent more frames than its share)
        uint64_t max_step = std::max<uint64_t>(1, std::min<uint64_t>(rq.max_batch, msb)) * R;
        // a piece is at least one whole generator chunk: for very short codes that is more frames than one launch takes, however
        // small the step (round-3 ADVICE) — such a code is turned away here, with a reason, on every rank alike; otherwise the
        // largest step is brought down to what the output buffers of one launch per rank hold
        while (max_step > static_cast<uint64_t>(R) && eng.shard_capacity(max_step, R) > eng.max_sub_batch())
            max_step = std::max<uint64_t>(R, max_step * 3 / 4);
        if (eng.shard_capacity(max_step, R) > eng.max_sub_batch())
            throw std::runtime_error("sharded simulation: one generator chunk of the noise stream holds more frames of this code than one "
                                     "launch takes (very short code): run it on one rank, or with a smaller LDPC_AMD_CHUNK_BLOCKS");
        const uint64_t min_step = std::min<uint64_t>(rq.first_batch, max_step);
        uint64_t step = min_step;
        bool go = true;
        while (go)
        {
            const uint64_t cap = eng.shard_capacity(step, R);
            it_buf.resize(cap), be_buf.resize(cap);
            BatchOut out;
            out.iters = it_buf.data(), out.bit_errors = be_buf.data();
            // A rank whose step fails (a HIP error, a device that went away) still takes part in the exchange below and says
            // so in the last word: every rank then leaves the loop with an error instead of waiting for the one that is gone.
            Engine::ShardStep st;
            std::string step_error;
            Fold mine;
            // (the encoder snapshot comes before the step's own exchange: if it fails here, the step is entered with the failure
            // in hand so that this rank still takes part in that exchange — round-3 ADVICE: it used to go straight to the
            // exchange below while the other ranks sat in the step's)
            std::string snap_error;
            try
            {
                eng.encoder_snapshot(nullptr);
            }
            catch (const std::exception &e)
            {
                snap_error = e.what();
            }
            try
            {
                st = eng.stream_decode_sharded(comm, rq.dec, step, out, nullptr, snap_error.empty() ? nullptr : &snap_error);
                // every rank's range as if all of it counted; the ranks before the one holding the stopping frame do
                mine = fold_range(it_buf.data(), be_buf.data(), st.n, 0, 0, kNoLimit, kNoLimit);
            }
            catch (const std::exception &e)
            {
                step_error = e.what();
            }
            uint64_t send[8] = {mine.n, mine.fec, mine.bec, mine.iters, mine.n_at_err, mine.iters_at_err,
                                static_cast<uint64_t>(*stop_flag ? 1 : 0), step_error.empty() ? 0u : 1u};
            std::vector<uint64_t> all(8 * static_cast<size_t>(R));
            comm.all_gather(send, all.data(), sizeof send);
            for (int q = 0; q < R; ++q)
                if (all[8 * static_cast<size_t>(q) + 7])
                    throw std::runtime_error(q == me ? "sharded simulation: " + step_error
                                                     : "sharded simulation: the step failed on rank " + std::to_string(q));
            bool any_stop_flag = false, err_seen = false;
            uint64_t err_frames = 0, err_iters = 0, used = 0;
            int q_stop = -1;
            for (int q = 0; q < R && q_stop < 0; ++q)
            {
                const uint64_t *L = &all[8 * static_cast<size_t>(q)];
                any_stop_flag = any_stop_flag || L[6];
                if (fec + L[1] >= rq.min_fec || frames + L[0] >= rq.max_frames)
                {
                    q_stop = q; // the stop rule fires inside (or at the end of) rank q's range
                    break;
                }
                if (L[1] > 0)
                    err_seen = true, err_frames = frames + L[4], err_iters = iters + L[5];
                frames += L[0], fec += L[1], bec += L[2], iters += L[3], used += L[0];
            }
            for (int q = 0; q < R; ++q)
                any_stop_flag = any_stop_flag || all[8 * static_cast<size_t>(q) + 6];
            uint64_t bec_at_err = bec;
            if (q_stop >= 0)
            {
                // the owner of the stopping frame walks its range again from the state the ranks before it leave
                Fold cut;
                if (q_stop == me)
                    cut = fold_range(it_buf.data(), be_buf.data(), st.n, frames, fec, rq.min_fec, rq.max_frames);
                uint64_t s2[8] = {cut.n, cut.fec, cut.bec, cut.iters, cut.n_at_err, cut.iters_at_err, cut.stopped, 0};
                comm.all_gather(s2, all.data(), sizeof s2);
                const uint64_t *C = &all[8 * static_cast<size_t>(q_stop)];
                if (C[1] > 0)
                    err_seen = true, err_frames = frames + C[4], err_iters = iters + C[5];
                frames += C[0], fec += C[1], bec += C[2], iters += C[3], used += C[0];
                bec_at_err = bec;
                go = false;
            }
            if (any_stop_flag)
                go = false;
            if (!go)
                eng.encoder_restore_and_skip(used, nullptr); // the encoder stands after the stopping frame, on every rank
            if (err_seen)
            {
                const uint64_t t_frame = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t_start).count()) /
                                         std::max<uint64_t>(err_frames, 1);
                const double fer = static_cast<double>(fec) / err_frames;
                const double ber = static_cast<double>(bec_at_err) / (err_frames * nc); // nc, not nct (ldpcsim.cpp:205)
                const double avg = static_cast<double>(err_iters) / err_frames;
                if (rq.cli_output && root)
                {
                    std::printf("\r %2lu/%2lu  |  %12lu  |  %.3f  |  %.2e  |  %.2e  |  %.1e  |  %.3fms", fec, rq.min_fec, err_frames,
                                xs[i], ber, fer, avg, static_cast<double>(t_frame) * 1e-3);
                    std::fflush(stdout);
                    char buf[160];
                    std::snprintf(buf, sizeof buf, "%lf %.3e %.3e %lu %.3e %.6f", xs[i], fer, ber, err_frames, avg,
                                  static_cast<double>(t_frame) * 1e-6);
                    lines[i + 1] = buf;
                    std::ofstream fp(rq.result_file);
                    if (fp.good())
                        for (const auto &l : lines)
                            fp << l << "\n";
                    else
                        std::printf("Warning: can not open logfile for writing\n");
                }
                if (results)
                {
                    results->fer[i] = fer, results->ber[i] = ber, results->avg_iter[i] = avg;
                    results->time[i] = static_cast<double>(t_frame) * 1e-6;
                    results->fec[i] = fec, results->frames[i] = err_frames;
                }
            }
            if (go)
            {
                uint64_t want = max_step;
                if (fec > 0)
                {
                    const double per_err = static_cast<double>(frames) / static_cast<double>(fec);
                    want = static_cast<uint64_t>(per_err * static_cast<double>(rq.min_fec - fec) * 1.25) + 1;
                }
                want = std::min<uint64_t>(want, rq.max_frames - frames);
                // a few step sizes only (powers of two between the smallest and the largest): every new size is a new piece
                // geometry — jump polynomials multiplied on the host, a table re-seek — inside the step (round-3 ADVICE)
                uint64_t q = min_step;
                while (q < want && q < max_step)
                    q = std::min<uint64_t>(q * 2, max_step);
                step = std::clamp<uint64_t>(q, min_step, max_step);
            }
        }
        if (rq.cli_output && root)
            std::printf("\n");
        if (totals)
        {
            totals[4 * i + 0] = frames, totals[4 * i + 1] = fec;
            totals[4 * i + 2] = bec, totals[4 * i + 3] = iters;
        }
    }
    return static_cast<int>(xs.size());
}
} // namespace

int run_simulation(Engine &eng, const SimRequest &rq, sim_results_t *results, uint64_t *totals, bool *stop_flag, Comm *comm)
{
    using clock = std::chrono::high_resolution_clock;
    static bool never_stop = false;
    if (!stop_flag)
        stop_flag = &never_stop;
    if (comm && comm->world() > 1)
        return run_simulation_sharded(eng, rq, results, totals, stop_flag, *comm);


    // channel points MIN, MIN+STEP, ... < MAX (ldpcsim.cpp:104-110); worst point first for BSC/BEC (:116-122)
    std::vector<double> xs;
    for (double v = rq.x_range[0]; v < rq.x_range[1]; v += rq.x_range[2])
        xs.push_back(v);
    const bool eps_axis = rq.channel == kBsc || rq.channel == kBec;
    if (eps_axis)
        std::reverse(xs.begin(), xs.end());

    std::vector<std::string> lines(xs.size() + 1);
    if (rq.cli_output)
        lines[0] = "snr fer ber frames avg_iter frame_time";

    std::cout << "=============================" << "===========================================================" << std::endl;
    std::cout << "  FEC   |      FRAME     |   " << (eps_axis ? "EPS" : "SNR")
              << "   |    BER     |    FER     | AVGITERS  |  TIME/FRAME   \n";
    std::cout << "========+================+===" << "======+============+============+===========+==============" << std::endl;

    const uint64_t nc = static_cast<uint64_t>(eng.code().nc());
    std::vector<uint32_t> it_buf, be_buf;
    for (size_t i = 0; i < xs.size(); ++i)
    {
        uint64_t bec = 0, fec = 0, frames = 0, iters = 0;
        auto t_start = clock::now();
        // the reference builds its channel objects once per run (ldpcsim.cpp:29-75): the info-word stream
        // and the accumulated codeword carry over from one channel point to the next
        eng.stream_begin(rq.channel, rq.seed, xs[i], /*fresh=*/i == 0);
        const uint64_t max_batch = std::min<uint64_t>(rq.max_batch, eng.max_sub_batch());
        uint64_t batch = std::min<uint64_t>(rq.first_batch, max_batch);
        bool go = true;
        while (go)
        {
            it_buf.resize(batch), be_buf.resize(batch);
            BatchOut out;
            out.iters = it_buf.data(), out.bit_errors = be_buf.data();
            eng.stream_decode(rq.dec, batch, out, nullptr);
            uint64_t used = 0;
            for (uint64_t f = 0; f < batch && go; ++f)
            {
                used = f + 1;
                iters += it_buf[f]; // accumulated for every decoded frame (ldpcsim.cpp:175-176)
                if (fec < rq.min_fec)
                {
                    ++frames;
                    if (be_buf[f] > 0)
                    {
                        auto t_now = clock::now();
                        uint64_t t_frame = static_cast<uint64_t>(
                            std::chrono::duration_cast<std::chrono::microseconds>(t_now - t_start).count());
                        t_frame /= frames;
                        bec += be_buf[f];
                        ++fec;
                        const double fer = static_cast<double>(fec) / frames;
                        const double ber = static_cast<double>(bec) / (frames * nc); // nc, not nct (ldpcsim.cpp:205)
                        const double avg = static_cast<double>(iters) / frames;
                        if (rq.cli_output)
                        {
                            std::printf("\r %2lu/%2lu  |  %12lu  |  %.3f  |  %.2e  |  %.2e  |  %.1e  |  %.3fms", fec,
                                        rq.min_fec, frames, xs[i], ber, fer, avg, static_cast<double>(t_frame) * 1e-3);
                            std::fflush(stdout);
                            char buf[160];
                            std::snprintf(buf, sizeof buf, "%lf %.3e %.3e %lu %.3e %.6f", xs[i], fer, ber, frames, avg,
                                          static_cast<double>(t_frame) * 1e-6);
                            lines[i + 1] = buf;
                            std::ofstream fp(rq.result_file);
                            if (fp.good())
                                for (const auto &l : lines)
                                    fp << l << "\n";
                            else
                                std::printf("Warning: can not open logfile for writing\n");
                        }
                        if (results)
                        {
                            results->fer[i] = fer;
                            results->ber[i] = ber;
                            results->avg_iter[i] = avg;
                            results->time[i] = static_cast<double>(t_frame) * 1e-6;
                            results->fec[i] = fec;
                            results->frames[i] = frames;
                        }
                        t_start += clock::now() - t_now; // printing is not charged to the frame time
                    }
                }
                go = fec < rq.min_fec && frames < rq.max_frames && !*stop_flag; // ldpcsim.cpp:255
            }
            if (!go) // frames decoded past the stopping frame never happened as far as the encoder is concerned
                eng.stream_rewind_encoder(batch - used, nullptr);
            // next batch: enough frames for the errors still missing at the observed rate, within bounds
            if (go)
            {
                uint64_t want = max_batch;
                if (fec > 0)
                {
                    double per_err = static_cast<double>(frames) / static_cast<double>(fec);
                    want = static_cast<uint64_t>(per_err * static_cast<double>(rq.min_fec - fec) * 1.25) + 1;
                }
                want = std::min<uint64_t>(want, rq.max_frames - frames);
                batch = std::clamp<uint64_t>(want, std::min(rq.first_batch, max_batch), max_batch);
            }
        }
        if (rq.cli_output)
            std::printf("\n");
        if (totals)
        {
            totals[4 * i + 0] = frames, totals[4 * i + 1] = fec;
            totals[4 * i + 2] = bec, totals[4 * i + 3] = iters;
        }
    }
    return static_cast<int>(xs.size());
}

} // namespace ldpc_amd
