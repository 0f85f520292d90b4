// mtstates.hpp — bookkeeping of the chunk start states of one mt19937_64 stream (host only, no HIP calls: unit-testable
// without a GPU through ldpc_hip_selftest_chunk_table / ldpc_hip_selftest_shard_table).
//
// The stream is cut into chunks of `chunk_blocks` twist blocks (312 words each).  A generator workgroup needs the
// 312-word state at the start of its chunk; states are obtained from one another by GF(2) jump-ahead
// (rng_kernels.hip, mt_jump_kernel: one task = one state advanced by a fixed polynomial t^(312 * chunk_blocks * s) mod phi,
// s = the stride in chunks).  Two access patterns, two tables:
//
//   StateRing     consecutive chunks (one rank reads the stream front to back: stream_decode, BSC/BEC raw words, the
//                 info-word stream).  Rows live in a ring of kRows rows, chunk c in row c % kRows.  The valid window
//                 [lo, hi) grows at its upper end: rows [hi, hi+n) = rows [hi-w, hi-w+n) advanced by w chunks, w the
//                 largest power of two <= min(hi - lo, kWindow), n <= w — ONE jump launch of n tasks per extension, n =
//                 the chunks the request consumes.  No doubling bursts: in steady state every request costs as many
//                 jump tasks as it has chunks.
//   StridedTable  a rank of a sharded stream reads `n` consecutive chunks per step and the steps are `stride` chunks
//                 apart (stride = world * piece): the table holds exactly those n rows and one in-place launch of n
//                 tasks with the single polynomial of `stride` chunks moves it to the next step — per-rank work is
//                 independent of the number of ranks.
//
// Both start from a SEEK: the state of an arbitrary chunk c, reached from chunk 0 (uploaded from the seed) or from a row
// the table already holds by one single-task jump per set bit of the distance.
#pragma once

#include <cstdint>
#include <vector>

#include "mt64.hpp"

namespace ldpc_amd
{

// t^(312 * chunk_blocks * n_chunks) mod phi, memoised per process (thread-safe)
const Gf2Poly &chunk_jump_poly(uint32_t chunk_blocks, uint64_t n_chunks);

struct StateOp
{
    enum Kind
    {
        kUpload0, // row dst := start state of chunk 0 (from the seed)
        kCopy,    // row dst := row src
        kJump     // rows (dst + i) % mod := rows (src + i) % mod advanced by `stride` chunks, i < n
    } kind;
    uint32_t src, dst, n, mod;
    uint64_t stride;
};

class StateRing
{
  public:
    static constexpr uint32_t kWindow = 2048;      // most chunks one request may span
    static constexpr uint32_t kRows = 2 * kWindow; // ring rows; two scratch rows follow (kRows, kRows + 1)
    static constexpr uint32_t kTotalRows = kRows + 2;
    void invalidate() { valid_ = false, req_lo_ = req_hi_ = 0; }
    // append the operations after which row c % kRows holds the start state of chunk c for every c in [c_lo, c_hi)
    void ensure(uint64_t c_lo, uint64_t c_hi, std::vector<StateOp> &ops);
    // look-ahead: append the operations that make the rows of the chunks up to c_hi valid as well (never at the expense of a
    // row the previous request reads; nothing when the ring holds no state yet).  The previous request stays what it was.
    void extend_to(uint64_t c_hi, std::vector<StateOp> &ops);
    uint64_t lo() const { return lo_; }
    uint64_t hi() const { return hi_; }
    bool valid() const { return valid_; }

  private:
    void seek(uint64_t c, std::vector<StateOp> &ops);
    void grow(uint64_t c_hi, std::vector<StateOp> &ops);
    bool valid_ = false;
    uint64_t lo_ = 0, hi_ = 0; // rows of chunks [lo_, hi_) are valid
    uint64_t req_lo_ = 0, req_hi_ = 0; // the previous request
};

class StridedTable
{
  public:
    static constexpr uint32_t kMaxRows = 4096;                       // rows of a table
    static constexpr uint32_t kSlots = 3;                            // tables: the one in use and up to two steps of look-ahead
    static constexpr uint32_t kTotalRows = kSlots * kMaxRows + 2;    // ... and two scratch rows
    void invalidate() { valid_ = false, ahead_ = 0; }
    // append the operations after which row base + i holds the start state of chunk first + i, i < n; returns base (a
    // multiple of kMaxRows).  Re-used when the geometry is unchanged and `first` is where the table stands or one stride
    // further: the step every rank takes costs ONE launch of n tasks from one table into the next — or nothing at all
    // when look_ahead() planned that launch already.
    uint32_t position(uint64_t first, uint32_t n, uint64_t stride, std::vector<StateOp> &ops);
    // look-ahead: append the launches (one per step, each into a table of its own) that bring the tables of the next TWO
    // steps into being: planned two steps before its table is read, a launch has a whole step to run beside the generator
    void look_ahead(std::vector<StateOp> &ops);
    uint32_t ahead() const { return ahead_; }
    uint64_t first() const { return first_; }
    uint32_t rows() const { return n_; }

  private:
    bool valid_ = false;
    uint32_t ahead_ = 0; // the next `ahead_` tables after the one in use hold (or are being given) the tables of the next steps
    uint32_t slot_ = 0;
    uint64_t first_ = 0, stride_ = 0;
    uint32_t n_ = 0;
};

} // namespace ldpc_amd
