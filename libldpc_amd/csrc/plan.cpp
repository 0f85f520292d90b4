// plan.cpp — builds the device execution plan (see plan.hpp).
#include "plan.hpp"

#include <algorithm>
#include <numeric>
#include <stdexcept>

namespace ldpc_amd
{

namespace
{
// longest-processing-time assignment of weighted blocks to waves
void deal(const std::vector<int> &cost, std::vector<uint16_t> &work, int &stride)
{
    std::vector<int> order(cost.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    std::vector<std::vector<uint16_t>> lists(kDecodeWaves);
    std::vector<long> load(kDecodeWaves, 0);
    for (int b : order)
    {
        int w = static_cast<int>(std::min_element(load.begin(), load.end()) - load.begin());
        lists[w].push_back(static_cast<uint16_t>(b));
        load[w] += cost[b];
    }
    stride = 1;
    for (auto &l : lists)
        stride = std::max(stride, static_cast<int>(l.size()));
    work.assign(static_cast<size_t>(kDecodeWaves) * stride, 0xFFFF);
    for (int w = 0; w < kDecodeWaves; ++w)
        std::copy(lists[w].begin(), lists[w].end(), work.begin() + static_cast<size_t>(w) * stride);
}
} // namespace

Plan build_plan(const LdpcCode &code)
{
    Plan p;
    const SparseGF2 &H = code.H;
    p.nc = code.nc(), p.mc = code.mc(), p.nnz = code.nnz(), p.nct = code.nct();
    p.max_cn_degree = code.max_cn_degree();
    p.max_vn_degree = code.max_vn_degree();

    // ---- check nodes: group by degree (ascending, stable), cut into wave-sized blocks ----
    std::vector<int> rows(p.mc);
    std::iota(rows.begin(), rows.end(), 0);
    auto rdeg = [&](int r) { return H.rptr[r + 1] - H.rptr[r]; };
    std::stable_sort(rows.begin(), rows.end(), [&](int a, int b) { return rdeg(a) < rdeg(b); });
    p.edge_slot.assign(p.nnz, 0);
    p.cn_rank_row.assign(rows.begin(), rows.end());
    uint32_t slot = 0;
    std::vector<int> cn_cost;
    for (int i = 0; i < p.mc;)
    {
        int d = rdeg(rows[i]), j = i;
        while (j < p.mc && j - i < kWaveSize && rdeg(rows[j]) == d)
            ++j;
        CnBlock b{slot, static_cast<uint16_t>(j - i), static_cast<uint16_t>(d)};
        for (int l = 0; l < j - i; ++l)
            for (int k = 0; k < d; ++k)
                p.edge_slot[H.redge[H.rptr[rows[i + l]] + k]] = slot + k * (j - i) + l;
        slot += static_cast<uint32_t>(d) * (j - i);
        p.cn_blocks.push_back(b);
        cn_cost.push_back(std::max(1, 3 * (d - 2)) * 8 + d);
        i = j;
    }

    // ---- variable nodes: group by degree (descending, stable) ----
    std::vector<int> cols(p.nc);
    std::iota(cols.begin(), cols.end(), 0);
    auto cdeg = [&](int c) { return H.cptr[c + 1] - H.cptr[c]; };
    std::stable_sort(cols.begin(), cols.end(), [&](int a, int b) { return cdeg(a) > cdeg(b); });
    p.rank_col.resize(p.nc), p.col_rank.resize(p.nc);
    for (int r = 0; r < p.nc; ++r)
    {
        p.rank_col[r] = static_cast<uint32_t>(cols[r]);
        p.col_rank[cols[r]] = static_cast<uint32_t>(r);
    }
    std::vector<int> vn_cost;
    for (int i = 0; i < p.nc;)
    {
        int d = cdeg(cols[i]), j = i;
        while (j < p.nc && j - i < kWaveSize && cdeg(cols[j]) == d)
            ++j;
        VnBlock b{static_cast<uint32_t>(p.vn_slot.size()), static_cast<uint32_t>(i), static_cast<uint16_t>(j - i),
                  static_cast<uint16_t>(d)};
        for (int k = 0; k < d; ++k)
            for (int l = 0; l < j - i; ++l)
                p.vn_slot.push_back(p.edge_slot[H.cedge[H.cptr[cols[i + l]] + k]]);
        p.vn_blocks.push_back(b);
        vn_cost.push_back(2 * d + 2);
        i = j;
    }
    p.rank_slot0.assign(p.nc, kNoSlot);
    for (int r = 0; r < p.nc; ++r)
        if (cdeg(cols[r]) > 0)
            p.rank_slot0[r] = p.edge_slot[H.cedge[H.cptr[cols[r]]]];
    if (p.vn_slot.empty())
        p.vn_slot.push_back(0);

    deal(cn_cost, p.cn_work, p.cn_work_stride);
    deal(vn_cost, p.vn_work, p.vn_work_stride);

    // ---- channel-side tables ----
    p.rank_kind.assign(p.nc, 0);
    for (int c : code.puncture)
        if (c >= 0 && c < p.nc)
            p.rank_kind[p.col_rank[c]] = 1;
    for (int c : code.shorten) // the reference applies shorten after puncture (channel.cpp:73-86)
        if (c >= 0 && c < p.nc)
            p.rank_kind[p.col_rank[c]] = 2;
    // nct = nc - |puncture| - |shorten| (ldpc.h:55) while bit_pos drops each listed column once: a column in
    // both lists leaves bit_pos longer than nct.  The channel then never writes the LLR of the surplus
    // positions (they keep the zero the decoder was constructed with) but the error count still visits them.
    p.n_bitpos = static_cast<int>(code.bit_pos.size());
    p.nct = std::min(p.nct, p.n_bitpos);
    p.tx_rank.resize(code.bit_pos.size());
    for (size_t i = 0; i < code.bit_pos.size(); ++i)
    {
        p.tx_rank[i] = p.col_rank[code.bit_pos[i]];
        if (static_cast<int>(i) >= p.nct)
            p.rank_kind[p.tx_rank[i]] = 3;
    }

    // ---- LDS footprint of one frame: messages (f64) + input LLRs (f64) + per-slot hard bits ----
    p.lds_bytes = static_cast<size_t>(8) * p.nnz + static_cast<size_t>(8) * p.nc + ((p.nnz + 15) / 16) * 16 + 16;
    p.lds_ok = code.min_cn_degree() >= 2 && p.max_cn_degree <= kMaxLdsCnDegree && p.lds_bytes <= 160 * 1024 &&
               p.cn_blocks.size() < 0xFFFF && p.vn_blocks.size() < 0xFFFF;
    p.hbm_ok = code.min_cn_degree() >= 2 && p.max_cn_degree <= kMaxCnDegree && p.cn_blocks.size() < 0xFFFF &&
               p.vn_blocks.size() < 0xFFFF;
    return p;
}

} // namespace ldpc_amd
