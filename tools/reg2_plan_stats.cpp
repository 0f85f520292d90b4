// Host-only check of the totals-form plan (plan.cpp, build_reg2_plan): LDS cycles per gather / scatter wave instruction
// implied by the placement, counted the way the LDS serves them (ds_read_b64: 32 lanes per cycle, bank = entry mod 32,
// equal addresses broadcast; ds_write_b64: 16 lanes per cycle, bank = entry mod 16).
//   g++ -O2 -std=c++20 -Ilibldpc_amd/csrc tools/reg2_plan_stats.cpp libldpc_amd/csrc/plan.cpp libldpc_amd/csrc/code.cpp -o /tmp/reg2_plan_stats
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

#include "code.hpp"
#include "plan.hpp"

using namespace ldpc_amd;

int main(int argc, char **argv)
{
    LdpcCode code(argv[1], "");
    Plan plan = build_plan(code);
    Reg2Plan r = build_reg2_plan(code, plan, 1024, 4, 6, 4, 4);
    if (!r.ok)
    {
        std::printf("plan refused\n");
        return 1;
    }
    const int NT = r.nt, S = r.kc * r.maxd;
    double g_cycles = 0, s_cycles[2] = {0, 0};
    long g_inst = 0, s_inst = 0;
    for (int s = 0; s < S; ++s)
        for (int w = 0; w < NT / 64; ++w)
        {
            ++g_inst, ++s_inst;
            for (int half = 0; half < 2; ++half)
            {
                std::map<uint32_t, std::vector<uint32_t>> bank;
                for (int l = 0; l < 32; ++l)
                {
                    const uint32_t e = (r.edge_w[static_cast<size_t>(s) * NT + w * 64 + half * 32 + l] & 0x3FFF8u) >> 3;
                    auto &v = bank[e % 32];
                    if (std::find(v.begin(), v.end(), e) == v.end())
                        v.push_back(e);
                }
                size_t worst = 0;
                for (auto &b : bank)
                    worst = std::max(worst, b.second.size());
                g_cycles += worst;
            }
            for (int rd = 0; rd < 2; ++rd)
                for (int q = 0; q < 4; ++q)
                {
                    int cnt[16] = {0};
                    bool trash_seen = false;
                    for (int l = 0; l < 16; ++l)
                    {
                        const uint32_t ew = r.edge_w[static_cast<size_t>(s) * NT + w * 64 + q * 16 + l];
                        uint32_t at = (((ew >> 15) | (ew << 17)) & 0x7FFF8u) - rd * 0x20000u;
                        const uint32_t trash = kReg2TrashEntry * 8u;
                        at = std::min(at, trash);
                        if (at == trash && trash_seen)
                            continue; // one address: one access
                        trash_seen |= at == trash;
                        ++cnt[(at >> 3) % 16];
                    }
                    s_cycles[rd] += *std::max_element(cnt, cnt + 16);
                }
        }
    // ---- invariants of the layout (plan.hpp): every mailbox entry below e_max is the target of at most one edge per
    // round, every edge of a node scatters into that node's round, gathers address a total or the neutral entry,
    // columns without an edge carry the "no edge" flag, everything fits 160 KB ----
    int bad = 0;
    {
        std::vector<int> hit[2] = {std::vector<int>(r.e_max, 0), std::vector<int>(r.e_max, 0)};
        size_t edges = 0;
        for (size_t i = 0; i < r.edge_w.size(); ++i)
        {
            const uint32_t ew = r.edge_w[i];
            const uint32_t gather = (ew & 0x3FFF8u) >> 3, scatter = ew >> 18, rd = ew & 1u;
            if (ew & 2u)
            {
                bad += gather != r.neutral;
                continue;
            }
            ++edges;
            bad += scatter >= r.e_max || gather >= r.lds_entries || gather == kReg2TrashEntry;
            if (scatter < r.e_max)
                ++hit[rd][scatter];
            // the address arithmetic of the kernel: own round -> the entry, other round -> the trash entry
            for (uint32_t round = 0; round < 2; ++round)
            {
                uint32_t at = (((ew >> 15) | (ew << 17)) & 0x7FFF8u) - round * 0x20000u;
                at = std::min(at, kReg2TrashEntry * 8u);
                bad += at != (round == rd ? scatter * 8u : kReg2TrashEntry * 8u);
            }
        }
        for (int rd = 0; rd < 2; ++rd)
            for (int h : hit[rd])
                bad += h > 1;
        bad += edges != static_cast<size_t>(plan.nnz);
        bad += static_cast<size_t>(r.lds_entries) * 8 + 64 > 160 * 1024 || r.neutral <= kReg2TrashEntry;
        // every variable-node block: its column and its total inside the regions, totals of different nodes distinct
        std::vector<int> tot_hit(r.lds_entries, 0);
        for (const Reg2VnBlock &b : r.vn_blocks)
            for (uint32_t l = 0; l < b.count; ++l)
            {
                bad += b.tot_off + l >= r.lds_entries || b.tot_off + l == kReg2TrashEntry || b.tot_off + l == r.neutral;
                if (b.tot_off + l < r.lds_entries)
                    bad += ++tot_hit[b.tot_off + l] > 1;
                bad += b.p0_off + l >= r.e_max;
                bad += b.degree > 1 && b.prest_off + static_cast<uint32_t>(b.degree - 2) * b.count + l >= r.e_max;
            }
    }
    std::printf("invariant violations: %d   regular-code instantiation: %s\n", bad, r.uniform_cn && r.uniform_vn ? "yes" : "no");
    std::printf("gather: %.2f LDS cycles per wave instruction (2 = no conflicts)\n", g_cycles / g_inst);
    std::printf("scatter: round 0 %.2f, round 1 %.2f LDS cycles per wave instruction (4 = no conflicts)\n", s_cycles[0] / s_inst,
                s_cycles[1] / s_inst);
    std::printf("lds entries %u (%u bytes), e_max %u, neutral %u\n", r.lds_entries, r.lds_entries * 8 + 16, r.e_max, r.neutral);
    return bad ? 2 : 0;
}
