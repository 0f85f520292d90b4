// mt64.cpp — see mt64.hpp.
#include "mt64.hpp"

#include <cstring>
#include <mutex>

namespace ldpc_amd
{

void mt64_next_window(uint64_t x[kMtWords])
{
    constexpr uint64_t UM = 0xFFFFFFFF80000000ull, LM = 0x7FFFFFFFull, A = 0xB5026F5AA96619E9ull;
    for (int k = 0; k < kMtWords; ++k)
    {
        uint64_t y = (x[k] & UM) | (x[(k + 1) % kMtWords] & LM);
        x[k] = x[(k + 156) % kMtWords] ^ (y >> 1) ^ ((y & 1) ? A : 0);
    }
}

void mt64_window0(uint64_t seed, uint64_t x[kMtWords])
{
    x[0] = seed;
    for (int i = 1; i < kMtWords; ++i)
        x[i] = 6364136223846793005ull * (x[i - 1] ^ (x[i - 1] >> 62)) + static_cast<uint64_t>(i);
    mt64_next_window(x); // the first draw of std::mt19937_64 triggers a full regeneration
}

uint64_t mt64_temper(uint64_t z)
{
    z ^= (z >> 29) & 0x5555555555555555ull;
    z ^= (z << 17) & 0x71D67FFFEDA60000ull;
    z ^= (z << 37) & 0xFFF7EEE000000000ull;
    z ^= (z >> 43);
    return z;
}

namespace
{
constexpr int kW = kMtWords + 1; // room for bit 19968+ during shifts

inline int get_bit(const uint64_t *p, int k) { return static_cast<int>(p[k >> 6] >> (k & 63) & 1); }

// dst ^= src << sh   (both n words)
void xor_shifted(uint64_t *dst, const uint64_t *src, int sh, int n)
{
    const int ws = sh >> 6, bs = sh & 63;
    for (int i = n - 1; i >= ws; --i)
    {
        uint64_t v = src[i - ws] << bs;
        if (bs && i - ws - 1 >= 0)
            v |= src[i - ws - 1] >> (64 - bs);
        dst[i] ^= v;
    }
}

// Berlekamp-Massey over GF(2) on one bit sequence of the generator
Gf2Poly compute_charpoly()
{
    const int N = 2 * kMtDegree + 64;
    // bit 63 of the untempered sequence words from an arbitrary non-degenerate seed
    std::vector<uint8_t> s(N);
    uint64_t win[kMtWords];
    mt64_window0(0x9E3779B97F4A7C15ull, win);
    for (int n = 0; n < N;)
    {
        for (int k = 0; k < kMtWords && n < N; ++k, ++n)
            s[n] = static_cast<uint8_t>(win[k] >> 63);
        mt64_next_window(win);
    }
    const int W = (kMtDegree + 2 + 63) / 64 + 1;
    std::vector<uint64_t> C(W, 0), B(W, 0), T(W), R(W, 0);
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < N; ++n)
    {
        // R bit i = s[n-i]
        for (int i = W - 1; i > 0; --i)
            R[i] = (R[i] << 1) | (R[i - 1] >> 63);
        R[0] = (R[0] << 1) | s[n];
        uint64_t acc = 0;
        const int lw = L / 64 + 1;
        for (int i = 0; i < lw && i < W; ++i)
            acc ^= C[i] & R[i];
        if (!(__builtin_popcountll(acc) & 1))
        {
            ++m;
            continue;
        }
        if (2 * L <= n)
        {
            T = C;
            xor_shifted(C.data(), B.data(), m, W);
            L = n + 1 - L;
            B = T;
            m = 1;
        }
        else
        {
            xor_shifted(C.data(), B.data(), m, W);
            ++m;
        }
    }
    // phi_k = C[L-k]
    Gf2Poly phi(kW, 0);
    if (L != kMtDegree)
        return Gf2Poly(); // cannot happen for mt19937_64; checked by the caller
    for (int k = 0; k <= L; ++k)
        if (get_bit(C.data(), L - k))
            phi[k >> 6] |= 1ull << (k & 63);
    return phi;
}

// r = r * t mod phi
inline void mul_t(uint64_t *r, const uint64_t *phi)
{
    for (int i = kW - 1; i > 0; --i)
        r[i] = (r[i] << 1) | (r[i - 1] >> 63);
    r[0] <<= 1;
    if (get_bit(r, kMtDegree))
        for (int i = 0; i < kW; ++i)
            r[i] ^= phi[i];
}
} // namespace

const Gf2Poly &mt64_charpoly()
{
    static Gf2Poly phi;
    static std::once_flag once;
    std::call_once(once, [] { phi = compute_charpoly(); });
    return phi;
}

Gf2Poly gf2_mulmod(const Gf2Poly &a, const Gf2Poly &b)
{
    const Gf2Poly &phi = mt64_charpoly();
    uint64_t r[kW] = {0};
    uint64_t aa[kW] = {0};
    std::memcpy(aa, a.data(), sizeof(uint64_t) * std::min<size_t>(a.size(), kW));
    for (int k = kMtDegree - 1; k >= 0; --k)
    {
        mul_t(r, phi.data());
        if (static_cast<size_t>(k >> 6) < b.size() && (b[k >> 6] >> (k & 63) & 1))
            for (int i = 0; i < kW; ++i)
                r[i] ^= aa[i];
    }
    return Gf2Poly(r, r + kMtWords);
}

Gf2Poly mt64_pow_t(uint64_t e)
{
    const Gf2Poly &phi = mt64_charpoly();
    Gf2Poly r(kMtWords, 0);
    r[0] = 1;
    bool started = false;
    for (int b = 63; b >= 0; --b)
    {
        if (started)
            r = gf2_mulmod(r, r);
        if (e >> b & 1)
        {
            uint64_t t[kW] = {0};
            std::memcpy(t, r.data(), sizeof(uint64_t) * kMtWords);
            mul_t(t, phi.data());
            r.assign(t, t + kMtWords);
            started = true;
        }
    }
    return r;
}

void mt64_jump_host(const uint64_t src[kMtWords], const Gf2Poly &g, uint64_t dst[kMtWords])
{
    std::vector<uint64_t> w(65 * kMtWords);
    uint64_t win[kMtWords];
    std::memcpy(win, src, sizeof win);
    for (int b = 0; b < 65; ++b)
    {
        std::memcpy(&w[static_cast<size_t>(b) * kMtWords], win, sizeof win);
        mt64_next_window(win);
    }
    for (int i = 0; i < kMtWords; ++i)
        dst[i] = 0;
    for (int k = 0; k < kMtDegree; ++k)
        if (g[k >> 6] >> (k & 63) & 1)
            for (int i = 0; i < kMtWords; ++i)
                dst[i] ^= w[k + i];
}

} // namespace ldpc_amd
