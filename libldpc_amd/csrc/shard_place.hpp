// shard_place.hpp — where the pieces of a sharded AWGN step lie in the one noise stream.
//
// A global step is world * m whole generator chunks of the raw mt19937_64 stream; rank q turns chunks [base + q m,
// base + (q+1) m) and a margin into accepted polar pairs.  ONE all-gather of three words per rank — pairs in the piece,
// pairs including the margin, a status word — and every rank knows where every piece starts in the pair sequence.  A frame
// (nct normals = nct / 2 pairs, libstdc++ normal_distribution: two normals per accepted pair) belongs to the rank whose
// piece holds its FIRST pair; its tail may reach into the margin.  Pure host arithmetic, no HIP: the engine calls it after
// the exchange, and the CPU tests call it through ldpc_hip_selftest_place over a real communicator.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace ldpc_amd
{

struct ShardPlacement
{
    uint64_t first = 0, n = 0;       // this rank's frames
    uint64_t step_frames = 0;        // frames of the whole step
    uint64_t pair_start = 0;         // stream index of the first pair of this rank's piece
    uint64_t pairs_after = 0;        // stream index of the first pair after the step
};

// all[3 q .. 3 q + 2] = rank q's {pairs in the piece, pairs including the margin, status (non-zero: the rank failed)}.
// pairs_before = stream index of the step's first pair, frame_pos = the step's first frame, cap = frames the output
// buffers hold.  Every rank evaluates every rank's conditions, so that all throw together (local_error: this rank's own
// failure text, if any).
inline ShardPlacement place_pieces(const uint64_t *all, int world, int rank, uint64_t pairs_before, uint64_t frame_pos, uint64_t nct,
                                   uint64_t cap, const std::string &local_error = std::string())
{
    for (int q = 0; q < world; ++q)
        if (all[3 * static_cast<size_t>(q) + 2])
            throw std::runtime_error(q == rank ? "sharded step failed on this rank: " + local_error
                                               : "sharded step failed on rank " + std::to_string(q));
    std::vector<uint64_t> P(static_cast<size_t>(world) + 1);
    P[0] = pairs_before;
    for (int q = 0; q < world; ++q)
        P[q + 1] = P[q] + all[3 * static_cast<size_t>(q)];
    auto first_frame = [&](uint64_t pair) { return (2 * pair + nct - 1) / nct; }; // first frame whose first pair is >= pair
    if (first_frame(P[0]) != frame_pos)
        throw std::runtime_error("sharded stream out of step");
    for (int q = 0; q < world; ++q) // the same checks on every rank, for every rank
    {
        const uint64_t f0 = first_frame(P[q]), nq = first_frame(P[q + 1]) - f0;
        if (nq > cap)
            throw std::runtime_error("sharded step: more frames in a piece than the output buffers hold");
        if (nq && ((f0 + nq) * nct - 1) / 2 - P[q] + 1 > all[3 * static_cast<size_t>(q) + 1])
            throw std::runtime_error("sharded step: a frame extends beyond the margin generated after the piece");
    }
    ShardPlacement s;
    s.first = first_frame(P[rank]);
    s.n = first_frame(P[rank + 1]) - s.first;
    s.step_frames = first_frame(P[world]) - frame_pos;
    s.pair_start = P[rank];
    s.pairs_after = P[world];
    return s;
}

} // namespace ldpc_amd
