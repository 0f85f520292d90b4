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
#include <string>
#include <vector>

namespace ldpc_amd
{

int run_simulation(Engine &eng, const SimRequest &rq, sim_results_t *results, uint64_t *totals, bool *stop_flag)
{
    using clock = std::chrono::high_resolution_clock;
    static bool never_stop = false;
    if (!stop_flag)
        stop_flag = &never_stop;

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
