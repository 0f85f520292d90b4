// mtstates.cpp — see mtstates.hpp.
#include "mtstates.hpp"

#include <algorithm>
#include <iterator>
#include <map>
#include <mutex>
#include <stdexcept>

namespace ldpc_amd
{

const Gf2Poly &chunk_jump_poly(uint32_t chunk_blocks, uint64_t n_chunks)
{
    static std::mutex mu;
    static std::map<std::pair<uint32_t, uint64_t>, Gf2Poly> cache;
    if (chunk_blocks == 0 || n_chunks == 0)
        throw std::runtime_error("mt19937_64 jump distance must be positive");
    if (mt64_charpoly().empty())
        throw std::runtime_error("mt19937_64 characteristic polynomial has unexpected degree");
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(chunk_blocks, n_chunks);
    auto it = cache.find(key);
    if (it != cache.end())
        return it->second;
    // powers of two by squaring (each from the one below, all memoised), anything else as their product
    auto pow2 = [&](unsigned m) -> const Gf2Poly & {
        for (unsigned k = 0; k <= m; ++k)
        {
            const auto kk = std::make_pair(chunk_blocks, uint64_t(1) << k);
            if (cache.count(kk))
                continue;
            if (k == 0)
                cache.emplace(kk, mt64_pow_t(static_cast<uint64_t>(kMtWords) * chunk_blocks));
            else
            {
                const Gf2Poly &h = cache.at(std::make_pair(chunk_blocks, uint64_t(1) << (k - 1)));
                cache.emplace(kk, gf2_mulmod(h, h));
            }
        }
        return cache.at(std::make_pair(chunk_blocks, uint64_t(1) << m));
    };
    // bounded: the powers of two (at most 64 per chunk size, what everything else is built from) stay; the products — one
    // per distinct step geometry a simulation ever used, 2.5 KB each — go when there are too many (callers copy what they get
    // before they ask again)
    if (cache.size() >= 256)
        for (auto e = cache.begin(); e != cache.end();)
            e = (e->first.second & (e->first.second - 1)) ? cache.erase(e) : std::next(e);
    Gf2Poly r;
    for (unsigned m = 0; m < 64; ++m)
        if (n_chunks >> m & 1)
        {
            const Gf2Poly &p = pow2(m);
            r = r.empty() ? p : gf2_mulmod(r, p);
        }
    return cache.emplace(key, std::move(r)).first->second;
}

namespace
{
// state of chunk `from_chunk + d` into scratch row a (or b), starting from the state in scratch row a: one single-task jump per
// set bit of d; returns the scratch row that holds the result
uint32_t jump_bits(uint64_t d, uint32_t a, uint32_t b, uint32_t mod, std::vector<StateOp> &ops)
{
    for (unsigned m = 0; m < 64; ++m)
        if (d >> m & 1)
        {
            ops.push_back({StateOp::kJump, a, b, 1, mod, uint64_t(1) << m});
            std::swap(a, b);
        }
    return a;
}
} // namespace

void StateRing::seek(uint64_t c, std::vector<StateOp> &ops)
{
    const uint32_t sa = kRows, sb = kRows + 1;
    uint64_t d = c;
    // from the newest row the ring holds when that is the shorter way
    if (valid_ && hi_ > lo_ && c >= hi_ - 1 && __builtin_popcountll(c - (hi_ - 1)) < __builtin_popcountll(c) + 1)
    {
        ops.push_back({StateOp::kCopy, static_cast<uint32_t>((hi_ - 1) % kRows), sa, 1, 0, 0});
        d = c - (hi_ - 1);
    }
    else
        ops.push_back({StateOp::kUpload0, 0, sa, 1, 0, 0});
    const uint32_t at = jump_bits(d, sa, sb, kTotalRows, ops);
    ops.push_back({StateOp::kCopy, at, static_cast<uint32_t>(c % kRows), 1, 0, 0});
    valid_ = true;
    lo_ = c, hi_ = c + 1;
}

void StateRing::ensure(uint64_t c_lo, uint64_t c_hi, std::vector<StateOp> &ops)
{
    if (c_hi <= c_lo)
        return;
    if (c_hi - c_lo > kWindow)
        throw std::runtime_error("mt19937_64 stream request exceeds the chunk-state window");
    const uint64_t n_req = c_hi - c_lo;
    if (valid_ && c_lo > hi_ && req_hi_ > req_lo_ && req_lo_ >= lo_ && req_hi_ <= hi_ && n_req <= req_hi_ - req_lo_)
    {
        // A request that lies AHEAD of the window by a gap (a rank of a sharded BSC / BEC stream reads its share of every
        // step: equal-sized requests a fixed distance apart): the rows of the previous request advanced by the distance
        // between the two, one launch of n tasks with that distance's polynomial — provided source and destination rows
        // do not overlap in the ring.
        const uint64_t dist = c_lo - req_lo_, dm = dist % kRows;
        if (dm >= n_req && dm <= kRows - n_req)
        {
            ops.push_back({StateOp::kJump, static_cast<uint32_t>(req_lo_ % kRows), static_cast<uint32_t>(c_lo % kRows),
                           static_cast<uint32_t>(n_req), kRows, dist});
            lo_ = c_lo, hi_ = c_hi;
        }
    }
    if (!valid_ || c_lo < lo_ || c_lo > hi_)
        seek(c_lo, ops);
    req_lo_ = c_lo, req_hi_ = c_hi;
    grow(c_hi, ops);
}

// rows of the chunks [hi_, c_hi): the newest w rows advanced by w chunks, w the largest power of two the ring holds
void StateRing::grow(uint64_t c_hi, std::vector<StateOp> &ops)
{
    while (hi_ < c_hi)
    {
        const uint64_t avail = std::min<uint64_t>(hi_ - lo_, kWindow);
        uint64_t w = 1;
        while (2 * w <= avail)
            w *= 2;
        const uint64_t n = std::min<uint64_t>(w, c_hi - hi_);
        ops.push_back({StateOp::kJump, static_cast<uint32_t>((hi_ - w) % kRows), static_cast<uint32_t>(hi_ % kRows),
                       static_cast<uint32_t>(n), kRows, w});
        hi_ += n;
        if (hi_ - lo_ > kRows)
            lo_ = hi_ - kRows;
    }
}

void StateRing::extend_to(uint64_t c_hi, std::vector<StateOp> &ops)
{
    if (!valid_ || hi_ <= lo_)
        return;
    if (req_hi_ > req_lo_) // (row c % kRows is overwritten by chunk c + kRows)
        c_hi = std::min(c_hi, req_lo_ + kRows);
    grow(c_hi, ops);
}

uint32_t StridedTable::position(uint64_t first, uint32_t n, uint64_t stride, std::vector<StateOp> &ops)
{
    if (n == 0 || n > kMaxRows || stride == 0)
        throw std::runtime_error("sharded mt19937_64 table: bad geometry");
    const uint32_t mod = kTotalRows, sa = kSlots * kMaxRows, sb = kSlots * kMaxRows + 1;
    if (valid_ && n == n_ && stride == stride_)
    {
        if (first == first_)
            return slot_ * kMaxRows;
        if (first == first_ + stride_) // the step every rank takes: n tasks, one polynomial, into the next table
        {
            const uint32_t next = (slot_ + 1) % kSlots;
            if (ahead_ == 0)
                ops.push_back({StateOp::kJump, slot_ * kMaxRows, next * kMaxRows, n, mod, stride});
            else
                --ahead_;
            slot_ = next;
            first_ = first;
            return slot_ * kMaxRows;
        }
    }
    ahead_ = 0; // (whatever the other tables hold is no longer what comes next)
    const uint32_t base = slot_ * kMaxRows;
    uint64_t d = first;
    if (valid_ && first >= first_ && __builtin_popcountll(first - first_) < __builtin_popcountll(first) + 1)
    {
        ops.push_back({StateOp::kCopy, base, sa, 1, 0, 0});
        d = first - first_;
    }
    else
        ops.push_back({StateOp::kUpload0, 0, sa, 1, 0, 0});
    const uint32_t at = jump_bits(d, sa, sb, mod, ops);
    ops.push_back({StateOp::kCopy, at, base, 1, 0, 0});
    for (uint32_t w = 1; w < n; w *= 2)
        ops.push_back({StateOp::kJump, base, base + w, std::min(w, n - w), mod, w});
    valid_ = true;
    first_ = first, n_ = n, stride_ = stride;
    return base;
}

void StridedTable::look_ahead(std::vector<StateOp> &ops)
{
    if (!valid_)
        return;
    while (ahead_ + 1 < kSlots) // (the table in use is never a destination)
    {
        const uint32_t src = (slot_ + ahead_) % kSlots, dst = (slot_ + ahead_ + 1) % kSlots;
        ops.push_back({StateOp::kJump, src * kMaxRows, dst * kMaxRows, n_, kTotalRows, stride_});
        ++ahead_;
    }
}

} // namespace ldpc_amd
