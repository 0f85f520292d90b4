// sim.hpp — batched Monte-Carlo loop (see sim.cpp).
#pragma once

#include <cstdint>
#include <string>

#include "../../include/ldpc_amd.h"
#include "engine.hpp"

namespace ldpc_amd
{

struct SimRequest
{
    DecParams dec;
    int channel = kAwgn;
    uint64_t seed = 0;
    double x_range[3] = {0, 0, 1};
    uint64_t max_frames = 10000000000ull;
    uint64_t min_fec = 50;
    std::string result_file;
    bool cli_output = false;       // per-error console line + result file (the reference's non-LIB_SHARED build)
    uint64_t first_batch = 4096;   // frames per launch before the error rate is known
    uint64_t max_batch = 65536;
};

// returns the number of channel points.  comm != nullptr with more than one rank: the frames of every step are shared
// out over the ranks (Engine::stream_decode_sharded) and the reference's counters and stop rule (ldpcsim.cpp:175-255)
// are reduced in rank order, so that every rank returns the counters of a one-rank run; rank 0 prints and writes.
int run_simulation(Engine &eng, const SimRequest &rq, sim_results_t *results, uint64_t *totals, bool *stop_flag,
                   Comm *comm = nullptr);

} // namespace ldpc_amd
