// plan.cpp — builds the device execution plan (see plan.hpp).
#include "plan.hpp"

#include "fused_rule.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>
#include <stdexcept>

namespace ldpc_amd
{

namespace
{
// longest-processing-time assignment of weighted blocks to waves
void deal(const std::vector<int> &cost, std::vector<uint32_t> &work, int &stride)
{
    std::vector<int> order(cost.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    std::vector<std::vector<uint32_t>> lists(kDecodeWaves);
    std::vector<long> load(kDecodeWaves, 0);
    for (int b : order)
    {
        int w = static_cast<int>(std::min_element(load.begin(), load.end()) - load.begin());
        lists[w].push_back(static_cast<uint32_t>(b));
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
        // relative issue cost of one block: d loads/stores + 3(d-2) box-pluses in the LLR-domain form, d divisions
        // (+ the partial results) in the likelihood-ratio form: both close to 10 d + 15 (d - 2) (measured with per-wave
        // phase timers on h.txt: with the old 8*3(d-2)+d weights the waves holding four degree-3 blocks and one
        // degree-4 block ran 10 % longer than those holding two of each)
        cn_cost.push_back(10 * d + 15 * (d - 2));
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
    int vn_cost_a = 2, vn_cost_b = 2; // per-edge and per-block share of a block's cost (LDPC_AMD_VN_COST=a,b: experiments)
    if (const char *e = std::getenv("LDPC_AMD_VN_COST"))
        std::sscanf(e, "%d,%d", &vn_cost_a, &vn_cost_b);
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
        vn_cost.push_back(vn_cost_a * d + vn_cost_b);
        i = j;
    }
    p.has_isolated_vn = p.nc > 0 && cdeg(cols[p.nc - 1]) == 0; // sorted by degree, descending
    p.rank_slot0.assign(p.nc, kNoSlot);
    for (int r = 0; r < p.nc; ++r)
        if (cdeg(cols[r]) > 0)
            p.rank_slot0[r] = p.edge_slot[H.cedge[H.cptr[cols[r]]]];
    if (p.vn_slot.empty())
        p.vn_slot.push_back(0);

    deal(cn_cost, p.cn_work, p.cn_work_stride);
    deal(vn_cost, p.vn_work, p.vn_work_stride);
    p.vn_work_desc.assign(static_cast<size_t>(kDecodeWaves) * (p.vn_work_stride + 1) * 4, 0);
    for (int w = 0; w < kDecodeWaves; ++w)
        for (int i = 0; i < p.vn_work_stride; ++i)
            if (uint32_t b = p.vn_work[static_cast<size_t>(w) * p.vn_work_stride + i]; b != 0xFFFF)
            {
                uint32_t *d = &p.vn_work_desc[(static_cast<size_t>(w) * (p.vn_work_stride + 1) + i) * 4];
                const VnBlock &vb = p.vn_blocks[b];
                d[0] = vb.idx_off, d[1] = vb.first, d[2] = vb.count | (static_cast<uint32_t>(vb.degree) << 16);
            }
    p.cn_desc_stride = (p.cn_work_stride + 1) / 2 * 2 + 2;
    p.cn_work_desc.assign(static_cast<size_t>(kDecodeWaves) * p.cn_desc_stride, CnBlock{0, 0, 0});
    for (int w = 0; w < kDecodeWaves; ++w)
        for (int i = 0; i < p.cn_work_stride; ++i)
            if (uint32_t b = p.cn_work[static_cast<size_t>(w) * p.cn_work_stride + i]; b != 0xFFFF)
                p.cn_work_desc[static_cast<size_t>(w) * p.cn_desc_stride + i] = p.cn_blocks[b];

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
    p.rank_kind.resize((static_cast<size_t>(p.nc) + 7) / 8 * 8, 0); // (the channel reads the kinds eight at a time)

    // ---- what a lane of the LDS-resident decoder keeps in registers, as it keeps it (plan.hpp, vn_packed) ----
    if (p.nnz < 65536 && p.vn_work_stride <= 8)
    {
        p.vn_packed.assign(static_cast<size_t>(kDecodeWaves) * kVnPackedRows * kWaveSize, 0);
        std::vector<uint32_t> rank_tx(p.nc, kVnSrcZero); // rank -> index among the transmitted bits
        for (int i = 0; i < p.nct; ++i)
            rank_tx[p.tx_rank[i]] = static_cast<uint32_t>(i);
        for (int w = 0; w < kDecodeWaves; ++w)
        {
            uint32_t *rows = &p.vn_packed[static_cast<size_t>(w) * kVnPackedRows * kWaveSize];
            std::fill(rows + 16 * kWaveSize, rows + 32 * kWaveSize, kVnSrcZero);
            for (int i = 0; i < p.vn_work_stride; ++i)
            {
                const uint32_t bi = p.vn_work[static_cast<size_t>(w) * p.vn_work_stride + i];
                if (bi == 0xFFFF)
                    continue;
                const VnBlock &vb = p.vn_blocks[bi];
                auto slot = [&](int edge, int lane) { return p.vn_slot[vb.idx_off + static_cast<size_t>(edge) * vb.count + lane]; };
                for (int l = 0; l < vb.count; ++l)
                {
                    if (vb.degree >= 1 && vb.degree <= 2)
                        rows[i * kWaveSize + l] = slot(0, l) | (slot(vb.degree - 1, l) << 16);
                    else if (i == 0 && vb.degree <= 16)
                        for (int q = 0; q < vb.degree; ++q)
                            rows[(8 + q / 2) * kWaveSize + l] |= slot(q, l) << (16 * (q & 1));
                    const uint32_t r = vb.first + l;
                    const uint8_t kind = p.rank_kind[r];
                    rows[(16 + i) * kWaveSize + l] = kind == 2 ? kVnSrcShortened : (kind == 0 ? rank_tx[r] : kVnSrcZero);
                    rows[(24 + i) * kWaveSize + l] = p.rank_col[r];
                }
            }
        }
    }

    // ---- LDS footprint of one frame: messages (f64) + input LLRs (f64) + per-slot hard bits ----
    p.lds_bytes = static_cast<size_t>(8) * p.nnz + static_cast<size_t>(8) * p.nc + ((p.nnz + 15) / 16) * 16 + 16;
    p.lds_ok = code.min_cn_degree() >= 2 && p.max_cn_degree <= kMaxLdsCnDegree && p.lds_bytes <= 160 * 1024 &&
               p.cn_blocks.size() < 0xFFFF && p.vn_blocks.size() < 0xFFFF;
    // the memory-resident decoder takes any check-node degree: up to kMaxCnDegree in registers, wider ones through a
    // scratch array (kernels.hip, cn_wide)
    p.hbm_ok = code.min_cn_degree() >= 2 && p.cn_blocks.size() < 0xFFFF && p.vn_blocks.size() < 0xFFFF;
    return p;
}

RegPlan build_reg_plan(const LdpcCode &code, const Plan &plan, int nt, int kc, int maxd, uint32_t lds_budget)
{
    RegPlan r;
    r.nt = nt, r.kc = kc, r.maxd = maxd;
    const int kRegWaves = nt / kWaveSize, kRegThreads = nt;
    const SparseGF2 &H = code.H;
    const int n_cb = static_cast<int>(plan.cn_blocks.size());
    if (code.min_cn_degree() < 2 || plan.max_cn_degree > maxd || n_cb > kc * kRegWaves || plan.nnz >= (1 << 28))
        return r;
    // mailbox: as many (double + hard-bit byte) entries as one CU's LDS holds
    const uint32_t cap = ((lds_budget - 256) / 9) & ~63u;
    // VN blocks (those of the LDS plan: sorted by degree, <= 64 equal-degree VNs) dealt to rounds in order
    std::vector<uint32_t> block_round(plan.vn_blocks.size()), block_off(plan.vn_blocks.size());
    uint32_t fill = 0, round = 0, peak = 0;
    r.round_first.push_back(0);
    for (size_t b = 0; b < plan.vn_blocks.size(); ++b)
    {
        const VnBlock &vb = plan.vn_blocks[b];
        const uint32_t need = static_cast<uint32_t>(vb.degree) * vb.count;
        if (need > cap)
            return r; // a single block of very high degree: not for this kernel
        if (fill + need > cap)
        {
            ++round;
            fill = 0;
            r.round_first.push_back(static_cast<uint32_t>(b));
        }
        block_round[b] = round;
        block_off[b] = fill;
        fill += need;
        peak = std::max(peak, fill);
        r.vn_blocks.push_back(RegVnBlock{vb.first, block_off[b], vb.count, vb.degree});
    }
    r.round_first.push_back(static_cast<uint32_t>(plan.vn_blocks.size()));
    r.rounds = static_cast<int>(round) + 1;
    if (r.rounds > 15)
        return r;
    r.mb_doubles = (peak + 63) & ~63u;

    // mailbox address of every edge: (VN block, position in the column, lane)
    std::vector<uint32_t> edge_mb(plan.nnz);
    for (size_t b = 0; b < plan.vn_blocks.size(); ++b)
    {
        const VnBlock &vb = plan.vn_blocks[b];
        for (int l = 0; l < vb.count; ++l)
        {
            const int col = static_cast<int>(plan.rank_col[vb.first + l]);
            for (int p = 0; p < vb.degree; ++p)
                edge_mb[H.cedge[H.cptr[col] + p]] = (block_round[b] << 28) | (block_off[b] + p * vb.count + l);
        }
    }
    r.cn_edge.assign(static_cast<size_t>(kc) * maxd * kRegThreads, kRegNoEdge);
    r.cn_deg.assign(static_cast<size_t>(kc) * kRegWaves, 0);
    r.cn_cnt.assign(static_cast<size_t>(kc) * kRegWaves, 0);
    int rank = 0; // CN rank in the LDS plan's order (cn_rank_row), block by block
    for (int bi = 0; bi < n_cb; ++bi)
    {
        const CnBlock &cb = plan.cn_blocks[bi];
        const int k = bi / kRegWaves, wave = bi % kRegWaves;
        r.cn_deg[k * kRegWaves + wave] = static_cast<uint8_t>(cb.degree);
        r.cn_cnt[k * kRegWaves + wave] = static_cast<uint8_t>(cb.count);
        for (int l = 0; l < cb.count; ++l, ++rank)
        {
            const int row = static_cast<int>(plan.cn_rank_row[rank]);
            const int tid = wave * kWaveSize + l;
            for (int j = 0; j < cb.degree; ++j)
                r.cn_edge[(static_cast<size_t>(k) * maxd + j) * kRegThreads + tid] = edge_mb[H.redge[H.rptr[row] + j]];
        }
    }
    r.ok = true;
    return r;
}

// Register-resident decoder, second form (plan.hpp).  VN blocks are the Plan's (<= 64 equal-degree nodes, degree
// descending); they are dealt to the two rounds so that both mailbox images have about the same number of entries.
Reg2Plan build_reg2_plan(const LdpcCode &code, const Plan &plan, int nt, int kc, int maxd, int nv0, int nv1)
{
    Reg2Plan r;
    r.nt = nt, r.kc = kc, r.maxd = maxd, r.nv0 = nv0, r.nv1 = nv1;
    const int W = nt / kWaveSize, NT = nt;
    const SparseGF2 &H = code.H;
    const int n_cb = static_cast<int>(plan.cn_blocks.size()), n_vb = static_cast<int>(plan.vn_blocks.size());
    if (code.min_cn_degree() < 2 || plan.max_cn_degree > maxd || n_cb > kc * W || plan.has_isolated_vn || kc * maxd > 64 ||
        n_vb > W * (nv0 + nv1) || plan.nnz == 0)
        return r;
    // a degree-1 variable node's v2c message is its channel ratio, not total x c2v (kernels.hip, vn_leaf_ratio): the totals
    // form cannot express that; such codes take the messages form
    for (const VnBlock &vb : plan.vn_blocks)
        if (vb.degree < 2)
            return r;
    const int cap[2] = {W * nv0, W * nv1};
    std::vector<int> rb[2]; // Plan VN-block ids per round
    long ent[2] = {0, 0};
    for (int b = 0; b < n_vb; ++b)
    {
        const bool ok0 = static_cast<int>(rb[0].size()) < cap[0], ok1 = static_cast<int>(rb[1].size()) < cap[1];
        const int rd = (ok0 && (!ok1 || ent[0] * cap[1] <= ent[1] * cap[0])) ? 0 : 1;
        rb[rd].push_back(b);
        ent[rd] += static_cast<long>(plan.vn_blocks[b].degree) * plan.vn_blocks[b].count;
    }
    // round 1: position-0 entries of all blocks first (64 per block: they receive the totals in place), then the rest
    const uint32_t c0_area = static_cast<uint32_t>(rb[1].size()) * kWaveSize;
    uint32_t e0 = 0, e1 = c0_area;
    r.vn_blocks.assign(static_cast<size_t>(W) * (nv0 + nv1), Reg2VnBlock{0, 0, 0, 0, 0});
    r.vn_rank.assign(static_cast<size_t>(W) * (nv0 + nv1) * kWaveSize, kNoSlot);
    std::vector<uint32_t> edge_gather(plan.nnz), edge_scatter(plan.nnz);
    std::vector<uint8_t> edge_round(plan.nnz);
    std::vector<Reg2VnBlock *> placed[2];
    std::vector<size_t> placed_pos[2];
    for (int rd = 0; rd < 2; ++rd)
        for (size_t q = 0; q < rb[rd].size(); ++q)
        {
            const VnBlock &vb = plan.vn_blocks[rb[rd][q]];
            const size_t pos = (q / W + (rd ? nv0 : 0)) * W + q % W; // (i, wave)
            Reg2VnBlock &o = r.vn_blocks[pos];
            o.count = vb.count, o.degree = vb.degree;
            if (rd == 0)
            {
                o.p0_off = e0, o.prest_off = e0 + vb.count;
                o.tot_off = static_cast<uint32_t>(q) * kWaveSize; // + e_max, below
                e0 += static_cast<uint32_t>(vb.degree) * vb.count;
            }
            else
            {
                o.p0_off = static_cast<uint32_t>(q) * kWaveSize, o.tot_off = o.p0_off;
                o.prest_off = e1;
                e1 += static_cast<uint32_t>(vb.degree - 1) * vb.count;
            }
            placed[rd].push_back(&o);
            placed_pos[rd].push_back(pos);
        }
    r.e_max = (std::max(e0, e1) + 31u) & ~31u;
    const uint32_t n_tot0 = static_cast<uint32_t>(rb[0].size()) * kWaveSize;
    // the trash entry sits at byte 0x20000 exactly (plan.hpp): the round-0 totals go below it when they fit there
    const uint32_t tot_base = r.e_max + n_tot0 <= kReg2TrashEntry ? r.e_max : kReg2TrashEntry + kWaveSize;
    r.neutral = std::max(tot_base + n_tot0, kReg2TrashEntry + 1);
    r.lds_entries = r.neutral + 2;
    if (r.e_max > kReg2TrashEntry || static_cast<size_t>(r.lds_entries) * 8 + 64 > 160 * 1024)
        return r;
    for (Reg2VnBlock *o : placed[0])
        o->tot_off += tot_base;

    // ---- which variable node sits in which (block, lane) ----
    // Every offset above is a multiple of 32 entries plus the lane (full blocks), so the LDS bank of a node's total
    // (gathered by the owners of its edges: ds_read_b64, 32 lanes per LDS cycle, bank = entry mod 32) and of its
    // column entries (scattered to: ds_write_b64, 16 lanes per cycle, bank = entry mod 16) is its lane mod 32 / mod 16.
    // The check-node side reads and writes in fixed lane groups — the j-th edges of 32 (16) neighbouring check nodes —
    // so nodes are placed greedily where their lane collides least with the nodes already placed in the same groups
    // (same degree class only: blocks are uniform in degree).  Natural order costs 3.5 LDS cycles per gather group
    // and 2.2 per scatter group on the (3,6) n=8192 code; this placement 2.0 and 1.15 (tools/reg2_plan_stats.cpp).  In the
    // round an edge's node does not belong to, the edge's owner writes the trash entry — one address for all lanes,
    // one more access to bank 0 of the group.
    struct Slot { uint32_t rd, q, lane; };
    const int max_deg = plan.max_vn_degree;
    std::vector<std::vector<std::vector<Slot>>> free_slots(max_deg + 1, std::vector<std::vector<Slot>>(64)); // [degree][rd*32 + colour]
    for (int rd = 0; rd < 2; ++rd)
        for (size_t q = rb[rd].size(); q-- > 0;)
        {
            const VnBlock &vb = plan.vn_blocks[rb[rd][q]];
            for (uint32_t l = vb.count; l-- > 0;)
                free_slots[vb.degree][rd * 32 + l % 32].push_back(Slot{static_cast<uint32_t>(rd), static_cast<uint32_t>(q), l});
        }
    // lane groups of the check-node side: edge -> (gather group, scatter group)
    const uint32_t n_gg = static_cast<uint32_t>(kc) * maxd * W * 2, n_sg = static_cast<uint32_t>(kc) * maxd * W * 4;
    std::vector<uint32_t> edge_gg(plan.nnz), edge_sg(plan.nnz);
    {
        int crank = 0;
        for (int bi = 0; bi < n_cb; ++bi)
        {
            const CnBlock &cb = plan.cn_blocks[bi];
            for (int l = 0; l < cb.count; ++l, ++crank)
            {
                const int row = static_cast<int>(plan.cn_rank_row[crank]);
                for (int j = 0; j < cb.degree; ++j)
                {
                    const int e = H.redge[H.rptr[row] + j];
                    const uint32_t inst = static_cast<uint32_t>(bi) * maxd + j; // one wave instruction
                    edge_gg[e] = inst * 2 + l / 32;
                    edge_sg[e] = inst * 4 + l / 16;
                }
            }
        }
    }
    std::vector<uint16_t> g_cnt(static_cast<size_t>(n_gg) * 32, 0), s_cnt(static_cast<size_t>(n_sg) * 2 * 16, 0);
    for (size_t g = 0; g < s_cnt.size(); g += 16)
        s_cnt[g + kReg2TrashEntry % 16] = 1; // the trash entry
    std::vector<uint32_t> order(plan.nc);
    std::iota(order.begin(), order.end(), 0u);
    uint64_t rng = 0x9E3779B97F4A7C15ull; // fixed seed: the plan is a pure function of the code
    for (size_t i = order.size(); i > 1; --i)
    {
        rng ^= rng >> 12, rng ^= rng << 25, rng ^= rng >> 27;
        std::swap(order[i - 1], order[(rng * 0x2545F4914F6CDD1Dull >> 33) % i]);
    }
    std::vector<Slot> rank_slot(plan.nc);
    for (uint32_t rank : order)
    {
        const int col = static_cast<int>(plan.rank_col[rank]);
        const int d = H.cptr[col + 1] - H.cptr[col];
        long best = -1;
        int best_class = -1;
        for (int cls = 0; cls < 64; ++cls)
        {
            if (free_slots[d][cls].empty())
                continue;
            const int rd = cls / 32, colour = cls % 32;
            long cost = 0;
            for (int p = 0; p < d; ++p)
            {
                const int e = H.cedge[H.cptr[col] + p];
                cost += 4l * g_cnt[static_cast<size_t>(edge_gg[e]) * 32 + colour] +
                        s_cnt[(static_cast<size_t>(edge_sg[e]) * 2 + rd) * 16 + colour % 16];
            }
            cost = cost * 4096 + static_cast<long>(4096 - free_slots[d][cls].size()); // ties: the emptiest class
            if (best < 0 || cost < best)
                best = cost, best_class = cls;
        }
        if (best_class < 0)
            return r; // cannot happen: the slots of a degree class equal its nodes
        const Slot sl = free_slots[d][best_class].back();
        free_slots[d][best_class].pop_back();
        rank_slot[rank] = sl;
        for (int p = 0; p < d; ++p)
        {
            const int e = H.cedge[H.cptr[col] + p];
            ++g_cnt[static_cast<size_t>(edge_gg[e]) * 32 + best_class % 32];
            ++s_cnt[(static_cast<size_t>(edge_sg[e]) * 2 + best_class / 32) * 16 + (best_class % 32) % 16];
        }
    }
    for (uint32_t rank = 0; rank < static_cast<uint32_t>(plan.nc); ++rank)
    {
        const Slot sl = rank_slot[rank];
        const Reg2VnBlock &o = *placed[sl.rd][sl.q];
        r.vn_rank[placed_pos[sl.rd][sl.q] * kWaveSize + sl.lane] = rank;
        const int col = static_cast<int>(plan.rank_col[rank]);
        const int d = H.cptr[col + 1] - H.cptr[col];
        for (int p = 0; p < d; ++p)
        {
            const int e = H.cedge[H.cptr[col] + p];
            edge_gather[e] = o.tot_off + sl.lane;
            edge_scatter[e] = p == 0 ? o.p0_off + sl.lane : o.prest_off + static_cast<uint32_t>(p - 1) * o.count + sl.lane;
            edge_round[e] = static_cast<uint8_t>(sl.rd);
        }
    }
    r.edge_w.assign(static_cast<size_t>(kc) * maxd * NT, (r.neutral << 3) | 2u);
    r.cn_deg.assign(static_cast<size_t>(kc) * W, 0), r.cn_cnt.assign(static_cast<size_t>(kc) * W, 0);
    int rank = 0; // CN rank in the Plan's order (cn_rank_row), block by block
    for (int bi = 0; bi < n_cb; ++bi)
    {
        const CnBlock &cb = plan.cn_blocks[bi];
        const int k = bi / W, wave = bi % W;
        r.cn_deg[k * W + wave] = static_cast<uint8_t>(cb.degree);
        r.cn_cnt[k * W + wave] = static_cast<uint8_t>(cb.count);
        for (int l = 0; l < cb.count; ++l, ++rank)
        {
            const int row = static_cast<int>(plan.cn_rank_row[rank]);
            const int tid = wave * kWaveSize + l;
            for (int j = 0; j < cb.degree; ++j)
            {
                const int e = H.redge[H.rptr[row] + j], s = k * maxd + j;
                r.edge_w[static_cast<size_t>(s) * NT + tid] = (edge_gather[e] << 3) | (edge_scatter[e] << 18) | edge_round[e];
            }
        }
    }
    r.uniform_cn = std::all_of(r.cn_deg.begin(), r.cn_deg.end(), [&](uint8_t d) { return d == maxd; });
    r.uniform_vn = nv0 >= 2 && nv1 >= 2;
    for (int rd = 0; rd < 2 && r.uniform_vn; ++rd)
    {
        const int nvr = rd ? nv1 : nv0;
        auto blk = [&](int q) -> const Reg2VnBlock & { return r.vn_blocks[static_cast<size_t>(q / W + (rd ? nv0 : 0)) * W + q % W]; };
        const Reg2VnBlock &b0 = blk(0), &b1 = blk(1);
        uint32_t *af = r.vn_affine[rd];
        af[0] = b0.p0_off, af[1] = b1.p0_off - b0.p0_off, af[2] = b0.prest_off, af[3] = b1.prest_off - b0.prest_off;
        af[4] = b0.tot_off, af[5] = b1.tot_off - b0.tot_off;
        for (int q = 0; q < nvr * W && r.uniform_vn; ++q)
        {
            const Reg2VnBlock &b = blk(q);
            r.uniform_vn = b.count == kWaveSize && b.degree == 3 && b.p0_off == af[0] + af[1] * q && b.prest_off == af[2] + af[3] * q &&
                           b.tot_off == af[4] + af[5] * q;
        }
    }
    r.ok = true;
    return r;
}

// Plan of the fused form (plan.hpp, detmath.h "Fused form"): check-node blocks by class with their inputs in the rule's
// order, message slots for the edges that do not end in a leaf, calls dealt to the waves by cost.
FusedPlan build_fused_plan(const LdpcCode &code, const Plan &plan)
{
    FusedPlan f;
    const SparseGF2 &H = code.H;
    if (!plan.lds_ok || plan.has_isolated_vn || H.rows <= 0 ||
        !dm_fused_applies(H.rows, H.cols, H.rptr.data(), H.rcol.data(), H.cptr.data()))
        return f;
    const int W = kDecodeWaves;
    auto cdeg = [&](int c) { return H.cptr[c + 1] - H.cptr[c]; };

    // ---- check nodes by class (key order), rows in file order inside a class, blocks of <= 64 ----
    struct Row
    {
        int row, order[4];
    };
    std::map<unsigned, std::vector<Row>> by_class;
    for (int i = 0; i < H.rows; ++i)
    {
        Row r{i, {0, 0, 0, 0}};
        unsigned flip;
        int leaf;
        const int d = H.rptr[i + 1] - H.rptr[i];
        if (!dm_fused_row_order(d, H.rcol.data() + H.rptr[i], H.cptr.data(), r.order, &flip, &leaf))
            return f;
        by_class[dm_fused_class(d, flip, leaf)].push_back(r);
    }
    struct Block
    {
        unsigned cls;
        uint32_t off; // first slot (in messages)
        std::vector<Row> rows;
    };
    std::vector<Block> blocks;
    f.edge_slot.assign(plan.nnz, kNoSlot);
    uint32_t slot = 0;
    for (auto &[cls, rows] : by_class)
    {
        const int d = static_cast<int>(cls & 7u), leaf = static_cast<int>((cls >> 3) & 1u), m = d - leaf;
        for (size_t i = 0; i < rows.size(); i += kWaveSize)
        {
            Block b{cls, slot, {}};
            b.rows.assign(rows.begin() + i, rows.begin() + std::min(rows.size(), i + kWaveSize));
            const uint32_t cnt = static_cast<uint32_t>(b.rows.size());
            for (uint32_t l = 0; l < cnt; ++l)
                for (int k = 0; k < m; ++k)
                    f.edge_slot[H.redge[H.rptr[b.rows[l].row] + b.rows[l].order[k]]] = slot + k * cnt + l;
            slot += static_cast<uint32_t>(m) * cnt;
            blocks.push_back(std::move(b));
        }
    }
    f.n_slots = static_cast<int>(slot);
    if (f.n_slots > DM_FUSED_MAX_SLOTS)
        return f;
    f.ho_map.assign(static_cast<size_t>(f.n_slots), 0);
    for (int e = 0; e < plan.nnz; ++e)
        if (f.edge_slot[e] != kNoSlot)
            f.ho_map[f.edge_slot[e]] = plan.edge_slot[e];

    // ---- calls: consecutive full blocks of a class two at a time ----
    struct Call
    {
        int b0, b1; // block ids (b1 = -1: one block)
        int cost;
        bool leaf;
    };
    auto class_cost = [](unsigned cls) {
        const int d = static_cast<int>(cls & 7u), leaf = static_cast<int>((cls >> 3) & 1u);
        return d == 2 ? 12 : (d == 3 ? (leaf ? 28 : 30) : (leaf ? 45 : 50));
    };
    std::vector<Call> calls;
    for (size_t b = 0; b < blocks.size();)
    {
        const bool pair = b + 1 < blocks.size() && blocks[b + 1].cls == blocks[b].cls && blocks[b].rows.size() == kWaveSize &&
                          blocks[b + 1].rows.size() == kWaveSize;
        const bool leaf = (blocks[b].cls >> 3) & 1u;
        calls.push_back(Call{static_cast<int>(b), pair ? static_cast<int>(b + 1) : -1, class_cost(blocks[b].cls) * (pair ? 2 : 1) + 4, leaf});
        b += pair ? 2 : 1;
    }
    // leaf calls first (each wave takes its share: they sit in register-backed slots), then the rest by cost
    std::vector<std::vector<int>> w_leaf(W), w_rest(W);
    std::vector<long> load(W, 0);
    {
        std::vector<int> order(calls.size());
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return calls[a].cost > calls[b].cost; });
        int n_leaf = 0;
        for (const Call &c : calls)
            n_leaf += c.leaf;
        const int leaf_cap = (n_leaf + W - 1) / W;
        if (leaf_cap > kFusedLeafCalls)
            return f;
        for (int ci : order)
            if (calls[ci].leaf)
            {
                int best = -1;
                for (int w = 0; w < W; ++w)
                    if (static_cast<int>(w_leaf[w].size()) < leaf_cap && (best < 0 || load[w] < load[best]))
                        best = w;
                w_leaf[best].push_back(ci);
                load[best] += calls[ci].cost;
            }
        for (int ci : order)
            if (!calls[ci].leaf)
            {
                const int w = static_cast<int>(std::min_element(load.begin(), load.end()) - load.begin());
                w_rest[w].push_back(ci);
                load[w] += calls[ci].cost;
            }
        f.cnl = leaf_cap;
    }
    auto describe = [&](const Call &c) {
        const Block &b0 = blocks[c.b0];
        FusedCall d{b0.off * 8u, static_cast<uint32_t>(b0.rows.size()), b0.cls, 0};
        if (c.b1 >= 0)
        {
            d.offs |= (blocks[c.b1].off * 8u) << 16;
            d.cnts |= static_cast<uint32_t>(blocks[c.b1].rows.size()) << 16;
        }
        return d;
    };
    f.leaf_calls.assign(static_cast<size_t>(W) * kFusedLeafCalls, FusedCall{0, 0, 0, 0});
    f.calls_stride = 1;
    for (int w = 0; w < W; ++w)
        f.calls_stride = std::max(f.calls_stride, static_cast<int>(w_rest[w].size()) + 1);
    f.calls.assign(static_cast<size_t>(W) * f.calls_stride, FusedCall{0, 0, 0, 0});
    for (int w = 0; w < W; ++w)
    {
        for (size_t i = 0; i < w_leaf[w].size(); ++i)
            f.leaf_calls[static_cast<size_t>(w) * kFusedLeafCalls + i] = describe(calls[w_leaf[w][i]]);
        for (size_t i = 0; i < w_rest[w].size(); ++i)
            f.calls[static_cast<size_t>(w) * f.calls_stride + i] = describe(calls[w_rest[w][i]]);
    }

    // ---- variable nodes of degree >= 2: by degree (descending), columns in file order, blocks of <= 64 ----
    std::vector<int> cols;
    for (int c = 0; c < H.cols; ++c)
        if (cdeg(c) >= 2)
            cols.push_back(c);
    std::stable_sort(cols.begin(), cols.end(), [&](int a, int b) { return cdeg(a) > cdeg(b); });
    struct VBlock
    {
        int degree;
        std::vector<int> cols;
    };
    std::vector<VBlock> vblocks;
    for (size_t i = 0; i < cols.size();)
    {
        size_t j = i;
        while (j < cols.size() && j - i < kWaveSize && cdeg(cols[j]) == cdeg(cols[i]))
            ++j;
        vblocks.push_back(VBlock{cdeg(cols[i]), std::vector<int>(cols.begin() + i, cols.begin() + j)});
        i = j;
    }
    std::vector<std::vector<int>> w_vn(W);
    {
        std::vector<int> order(vblocks.size());
        std::iota(order.begin(), order.end(), 0);
        auto cost = [&](int b) { return vblocks[b].degree == 2 ? 13 : 5 * vblocks[b].degree + 8; };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
        std::vector<long> vload(W, 0);
        for (int b : order)
        {
            int best = -1;
            for (int w = 0; w < W; ++w) // (one slot kept free for the alignment of the degree-2 blocks)
                if (static_cast<int>(w_vn[w].size()) < kFusedVnSlots - 1 && (best < 0 || vload[w] < vload[best]))
                    best = w;
            if (best < 0)
                for (int w = 0; w < W; ++w)
                    if (static_cast<int>(w_vn[w].size()) < kFusedVnSlots && (best < 0 || vload[w] < vload[best]))
                        best = w;
            if (best < 0)
                return f;
            w_vn[best].push_back(b);
            vload[best] += cost(b);
        }
    }
    // entry of a column's channel value in the staging area, and the flags of its column word
    std::vector<uint32_t> col_entry(H.cols, static_cast<uint32_t>(plan.nct) + 1u), col_word(H.cols);
    for (int c = 0; c < H.cols; ++c)
        col_word[c] = static_cast<uint32_t>(c);
    for (int i = 0; i < static_cast<int>(code.bit_pos.size()); ++i)
    {
        if (i < plan.nct)
            col_entry[code.bit_pos[i]] = static_cast<uint32_t>(i);
        col_word[code.bit_pos[i]] |= kFusedCounted;
    }
    for (int c : code.shorten) // (after the transmitted ones: a column in both lists is not in bit_pos)
        if (c >= 0 && c < H.cols && plan.rank_kind[plan.col_rank[c]] == 2)
            col_entry[c] = static_cast<uint32_t>(plan.nct);
    f.has_shortened = !code.shorten.empty();
    for (int c = 0; c < H.cols; ++c) // (punctured and never-written columns have L = 0: lambda = rho = 1, staged as constants)
        if (cdeg(c) >= 3 && col_entry[c] <= static_cast<uint32_t>(plan.nct))
            f.need_lambda = true;

    f.vn_desc.assign(static_cast<size_t>(W) * kFusedVnSlots * 4, 0);
    f.lane_tab.assign(static_cast<size_t>(W) * kFusedLaneRows * kWaveSize, 0);
    f.vnb = 1;
    for (int w = 0; w < W; ++w)
    {
        uint32_t *tab = &f.lane_tab[static_cast<size_t>(w) * kFusedLaneRows * kWaveSize];
        for (int r = 16; r < 24; ++r)
            std::fill(tab + r * kWaveSize, tab + (r + 1) * kWaveSize, (static_cast<uint32_t>(plan.nct) + 1u) * 16u);
        for (int r = 32; r < 36; ++r)
            std::fill(tab + r * kWaveSize, tab + (r + 1) * kWaveSize, (static_cast<uint32_t>(plan.nct) + 1u) * 16u);
        for (int r = 24; r < 32; ++r)
            std::fill(tab + r * kWaveSize, tab + (r + 1) * kWaveSize, kFusedNone);
        for (int r = 36; r < 40; ++r)
            std::fill(tab + r * kWaveSize, tab + (r + 1) * kWaveSize, kFusedNone);
        for (int r = 40; r < 44; ++r)
            std::fill(tab + r * kWaveSize, tab + (r + 1) * kWaveSize, 0u);
        // widest first; the degree-2 blocks from an even slot on
        std::vector<int> &mine = w_vn[w];
        std::stable_sort(mine.begin(), mine.end(), [&](int a, int b) { return vblocks[a].degree > vblocks[b].degree; });
        std::vector<int> slots;
        for (int b : mine)
        {
            if (vblocks[b].degree == 2 && slots.size() % 2 == 1 && !slots.empty() && slots.back() >= 0 && vblocks[slots.back()].degree != 2 &&
                mine.size() + 1 <= static_cast<size_t>(kFusedVnSlots))
                slots.push_back(-1);
            slots.push_back(b);
        }
        if (static_cast<int>(slots.size()) > kFusedVnSlots)
            return f;
        f.vnb = std::max(f.vnb, static_cast<int>(slots.size()));
        for (size_t sw = 0; sw < slots.size(); ++sw)
        {
            if (slots[sw] < 0)
                continue;
            const VBlock &vb = vblocks[slots[sw]];
            const uint32_t cnt = static_cast<uint32_t>(vb.cols.size());
            uint32_t *d = &f.vn_desc[(static_cast<size_t>(w) * kFusedVnSlots + sw) * 4];
            d[0] = cnt | (static_cast<uint32_t>(vb.degree) << 16);
            d[1] = static_cast<uint32_t>(f.vn_slot.size());
            auto eslot = [&](int l, int p) { return f.edge_slot[H.cedge[H.cptr[vb.cols[l]] + p]] * 8u; };
            const bool in_regs = vb.degree == 2 || (sw == 0 && vb.degree <= 15);
            f.vn_prog[w] |= (vb.degree == 2 ? (cnt == kWaveSize ? kFusedVnPair : kFusedVn2) : (in_regs ? kFusedVnWide : kFusedVnTable)) << (4 * sw);
            if (!in_regs)
                for (int p = 0; p < vb.degree; ++p)
                    for (uint32_t l = 0; l < cnt; ++l)
                        f.vn_slot.push_back(eslot(static_cast<int>(l), p));
            for (uint32_t l = 0; l < cnt; ++l)
            {
                if (vb.degree == 2)
                    tab[sw * kWaveSize + l] = eslot(l, 0) | (eslot(l, 1) << 16);
                else if (in_regs)
                    for (int q = 0; q < vb.degree; ++q)
                        tab[(8 + q / 2) * kWaveSize + l] |= eslot(l, q) << (16 * (q & 1));
                tab[(16 + sw) * kWaveSize + l] = col_entry[vb.cols[l]] * 16u;
                tab[(24 + sw) * kWaveSize + l] = col_word[vb.cols[l]];
            }
        }
        for (size_t c = 0; c < w_leaf[w].size(); ++c)
            for (int h = 0; h < 2; ++h)
            {
                const int bi = h == 0 ? calls[w_leaf[w][c]].b0 : calls[w_leaf[w][c]].b1;
                if (bi < 0)
                    continue;
                const Block &b = blocks[bi];
                const int d = static_cast<int>(b.cls & 7u);
                for (size_t l = 0; l < b.rows.size(); ++l)
                {
                    const int col = H.rcol[H.rptr[b.rows[l].row] + b.rows[l].order[d - 1]]; // the leaf: the last input
                    tab[(32 + 2 * c + h) * kWaveSize + l] = col_entry[col] * 16u;
                    tab[(36 + 2 * c + h) * kWaveSize + l] = col_word[col];
                    tab[(40 + 2 * c + h) * kWaveSize + l] = plan.edge_slot[H.redge[H.rptr[b.rows[l].row] + b.rows[l].order[d - 1]]];
                }
            }
    }
    if (f.vn_slot.empty())
        f.vn_slot.push_back(0);
    f.wide_exclusive = true;
    for (int w = 0; w < W; ++w)
    {
        if ((f.vn_prog[w] & 0xFu) == kFusedVnWide && (f.vn_prog[w] >> 4) != 0)
            f.wide_exclusive = false;
        for (int sw = 0; sw < kFusedVnSlots; ++sw) // (and no block that goes through the slot table)
            if (((f.vn_prog[w] >> (4 * sw)) & 0xFu) == kFusedVnTable)
                f.wide_exclusive = false;
    }
    f.ok = true;
    return f;
}

// Steps of the layered schedule: check nodes in file order within each degree, each put into the first step of its degree
// that has a free lane and none of its variable nodes yet (greedy; a variable node of degree d forces its d check nodes
// into d different steps).  Steps are ordered by the file position of their first check node, so a sweep still walks H
// roughly top to bottom.
LayerPlan build_layer_plan(const LdpcCode &code, const Plan &plan)
{
    LayerPlan L;
    const SparseGF2 &H = code.H;
    if (H.cols > 0xFFFF || code.max_cn_degree() > kMaxLdsCnDegree || code.min_cn_degree() < 2)
        return L;
    struct Open
    {
        int degree;
        std::vector<int> rows;
        std::vector<uint8_t> used; // per column
    };
    std::vector<Open> open;
    for (int i = 0; i < H.rows; ++i)
    {
        const int deg = H.rptr[i + 1] - H.rptr[i];
        Open *dst = nullptr;
        for (Open &o : open)
        {
            if (o.degree != deg || static_cast<int>(o.rows.size()) >= kWaveSize)
                continue;
            bool clash = false;
            for (int p = H.rptr[i]; p < H.rptr[i + 1] && !clash; ++p)
                clash = o.used[H.rcol[p]] != 0;
            if (!clash)
            {
                dst = &o;
                break;
            }
        }
        if (!dst)
        {
            open.push_back({deg, {}, std::vector<uint8_t>(H.cols, 0)});
            dst = &open.back();
        }
        dst->rows.push_back(i);
        for (int p = H.rptr[i]; p < H.rptr[i + 1]; ++p)
            dst->used[H.rcol[p]] = 1;
    }
    L.step_of_row.assign(H.rows, -1);
    for (size_t si = 0; si < open.size(); ++si)
        for (int i : open[si].rows)
            L.step_of_row[i] = static_cast<int>(si);
    for (const Open &o : open)
    {
        LayerStep st{static_cast<uint32_t>(L.vn.size()), static_cast<uint16_t>(o.rows.size()), static_cast<uint16_t>(o.degree)};
        L.vn.resize(L.vn.size() + static_cast<size_t>(o.degree) * kWaveSize, 0);
        for (size_t l = 0; l < o.rows.size(); ++l)
        {
            const int i = o.rows[l];
            for (int j = 0; j < o.degree; ++j)
                L.vn[st.off + static_cast<size_t>(j) * kWaveSize + l] = static_cast<uint16_t>(plan.col_rank[H.rcol[H.rptr[i] + j]]);
        }
        L.steps.push_back(st);
    }
    L.slots = static_cast<uint32_t>(L.vn.size());
    L.ok = !L.steps.empty();
    return L;
}

} // namespace ldpc_amd
