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
    std::printf("gather: %.2f LDS cycles per wave instruction (2 = no conflicts)\n", g_cycles / g_inst);
    std::printf("scatter: round 0 %.2f, round 1 %.2f LDS cycles per wave instruction (4 = no conflicts)\n", s_cycles[0] / s_inst,
                s_cycles[1] / s_inst);
    std::printf("lds entries %u (%u bytes), e_max %u, neutral %u\n", r.lds_entries, r.lds_entries * 8 + 16, r.e_max, r.neutral);
    return 0;
}
