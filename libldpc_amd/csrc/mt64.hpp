// mt64.hpp — host side of the mt19937_64 stream engine: seeding, and the GF(2) polynomial algebra
// behind jump-ahead (characteristic polynomial by Berlekamp-Massey, t^e mod phi(t)).
//
// The sequence s_n is the n-th 64-bit word the generator produces BEFORE tempering; the n-th output
// of std::mt19937_64(seed) is temper(s_n).  A "window" at position n is (s_n .. s_{n+311}).
// Every bit position of the sequence obeys one linear recurrence over GF(2) whose characteristic
// polynomial phi has degree 19937, hence for g(t) = t^J mod phi:  s_{n+J+i} = XOR_{k: g_k=1} s_{n+k+i}
// (the 31 low bits of the first word of a jumped window are never used by the generator).
#pragma once

#include <cstdint>
#include <vector>

namespace ldpc_amd
{

constexpr int kMtWords = 312;
constexpr int kMtDegree = 19937;

using Gf2Poly = std::vector<uint64_t>; // coefficient k at word k/64, bit k%64

// window at position 0 for std::mt19937_64(seed) (libstdc++ bits/random.tcc seed + first twist)
void mt64_window0(uint64_t seed, uint64_t window[kMtWords]);
// advance a window by 312 positions in place
void mt64_next_window(uint64_t window[kMtWords]);
uint64_t mt64_temper(uint64_t z);

// characteristic polynomial of the recurrence (degree 19937, computed once, cached)
const Gf2Poly &mt64_charpoly();
// t^e mod phi (312 words)
Gf2Poly mt64_pow_t(uint64_t e);
Gf2Poly gf2_mulmod(const Gf2Poly &a, const Gf2Poly &b);
// host evaluation of the jump (tests; the product path runs mt_jump_kernel)
void mt64_jump_host(const uint64_t src[kMtWords], const Gf2Poly &g, uint64_t dst[kMtWords]);

} // namespace ldpc_amd
