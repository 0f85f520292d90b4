// code.cpp — code-file loader (reference behaviour: src/core/ldpc.cpp:40-101, src/core/sparse.h:92-153).
#include "code.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace ldpc_amd
{

static bool blank(const std::string &s)
{
    return s.find_first_not_of(" \t\r\n") == std::string::npos;
}

void SparseGF2::read(const std::string &path, int skip_lines)
{
    std::ifstream in(path);
    if (!in.good())
        throw std::runtime_error("can not open file for reading");
    std::string line;
    while (skip_lines-- > 0)
        std::getline(in, line);
    erow.clear();
    ecol.clear();
    int maxr = 0, maxc = 0;
    while (std::getline(in, line))
    {
        // "row col [value]"; every stored entry is 1 over GF(2) (a missing or zero value becomes 1,
        // sparse.h:126-130).  A blank line is undefined behaviour in the reference; it is skipped here.
        if (blank(line))
            continue;
        const char *p = line.c_str();
        char *end = nullptr;
        long r = std::strtol(p, &end, 10);
        long c = std::strtol(end, &end, 10);
        erow.push_back(static_cast<int>(r));
        ecol.push_back(static_cast<int>(c));
        maxr = std::max(maxr, static_cast<int>(r));
        maxc = std::max(maxc, static_cast<int>(c));
    }
    rows = maxr + 1; // dimensions = max index + 1 (sparse.h:136-143)
    cols = maxc + 1;
    build_adjacency();
}

void SparseGF2::build_adjacency()
{
    const int n = nnz();
    rptr.assign(rows + 1, 0);
    cptr.assign(cols + 1, 0);
    for (int e = 0; e < n; ++e)
    {
        if (erow[e] < 0 || ecol[e] < 0)
            throw std::runtime_error("negative index in matrix file");
        ++rptr[erow[e] + 1];
        ++cptr[ecol[e] + 1];
    }
    for (int i = 0; i < rows; ++i)
        rptr[i + 1] += rptr[i];
    for (int j = 0; j < cols; ++j)
        cptr[j + 1] += cptr[j];
    rcol.resize(n), redge.resize(n), crow.resize(n), cedge.resize(n);
    std::vector<int> rfill(rptr.begin(), rptr.end() - 1), cfill(cptr.begin(), cptr.end() - 1);
    for (int e = 0; e < n; ++e) // stable fill keeps file order per node
    {
        int pr = rfill[erow[e]]++, pc = cfill[ecol[e]]++;
        rcol[pr] = ecol[e], redge[pr] = e;
        crow[pc] = erow[e], cedge[pc] = e;
    }
}

void SparseGF2::multiply_right(const uint8_t *right, uint8_t *result) const
{
    for (int i = 0; i < rows; ++i)
    {
        uint8_t s = result[i];
        for (int p = rptr[i]; p < rptr[i + 1]; ++p)
            s ^= static_cast<uint8_t>(right[rcol[p]] != 0);
        result[i] = s;
    }
}

void SparseGF2::multiply_left(const uint8_t *left, uint8_t *result) const
{
    for (int j = 0; j < cols; ++j)
    {
        uint8_t s = result[j];
        for (int p = cptr[j]; p < cptr[j + 1]; ++p)
            s ^= static_cast<uint8_t>(left[crow[p]] != 0);
        result[j] = s;
    }
}

// GF(2) rank by bit-packed Gaussian elimination (cold path; value equals the reference's
// list-based elimination sparse.h:233-300, e.g. 1021 for tests/code/h.txt).
int SparseGF2::rank() const
{
    const int words = (cols + 63) / 64;
    std::vector<uint64_t> a(static_cast<size_t>(rows) * words, 0);
    for (int e = 0; e < nnz(); ++e)
        a[static_cast<size_t>(erow[e]) * words + ecol[e] / 64] ^= 1ull << (ecol[e] % 64);
    int rk = 0;
    for (int c = 0; c < cols && rk < rows; ++c)
    {
        int piv = -1;
        for (int r = rk; r < rows && piv < 0; ++r)
            if (a[static_cast<size_t>(r) * words + c / 64] >> (c % 64) & 1)
                piv = r;
        if (piv < 0)
            continue;
        if (piv != rk)
            std::swap_ranges(a.begin() + static_cast<size_t>(piv) * words,
                             a.begin() + static_cast<size_t>(piv + 1) * words, a.begin() + static_cast<size_t>(rk) * words);
        for (int r = rk + 1; r < rows; ++r)
            if (a[static_cast<size_t>(r) * words + c / 64] >> (c % 64) & 1)
                for (int w = c / 64; w < words; ++w)
                    a[static_cast<size_t>(r) * words + w] ^= a[static_cast<size_t>(rk) * words + w];
        ++rk;
    }
    return rk;
}

LdpcCode::LdpcCode(const std::string &pc_file, const std::string &gen_file)
{
    std::ifstream in(pc_file);
    if (!in.good())
        throw std::runtime_error("can not open file for reading");
    // Leading lines containing ':' are the legacy header; only puncture / shorten lists matter
    // (ldpc.cpp:50-80).
    std::string line;
    int skip = 0;
    while (std::getline(in, line))
    {
        auto pos = line.find(':');
        if (pos == std::string::npos)
            break;
        std::string token = line.substr(0, pos);
        std::istringstream rec(line.substr(pos + 1));
        int idx;
        if (token.find("puncture") != std::string::npos)
            while (rec >> idx)
                puncture.push_back(idx);
        else if (token.find("shorten") != std::string::npos)
            while (rec >> idx)
                shorten.push_back(idx);
        ++skip;
    }
    in.close();
    H.read(pc_file, skip);
    max_degree = std::max(max_cn_degree(), max_vn_degree());
    for (int i = 0; i < nc(); ++i) // ldpc.cpp:90-100
    {
        if (std::find(shorten.begin(), shorten.end(), i) != shorten.end())
            continue;
        if (std::find(puncture.begin(), puncture.end(), i) != puncture.end())
            continue;
        bit_pos.push_back(i);
    }
    if (!gen_file.empty())
        G.read(gen_file, 0);
}

int LdpcCode::min_cn_degree() const
{
    int d = H.rows ? H.nnz() : 0;
    for (int i = 0; i < H.rows; ++i)
        d = std::min(d, H.rptr[i + 1] - H.rptr[i]);
    return d;
}
int LdpcCode::max_cn_degree() const
{
    int d = 0;
    for (int i = 0; i < H.rows; ++i)
        d = std::max(d, H.rptr[i + 1] - H.rptr[i]);
    return d;
}
int LdpcCode::max_vn_degree() const
{
    int d = 0;
    for (int j = 0; j < H.cols; ++j)
        d = std::max(d, H.cptr[j + 1] - H.cptr[j]);
    return d;
}

static void print_list(std::ostringstream &os, const std::vector<int> &v)
{
    os << "[";
    for (size_t i = 0; i < v.size(); ++i)
        os << v[i] << (i + 1 < v.size() ? ", " : "");
    os << "]";
}

std::string LdpcCode::describe() const
{
    std::ostringstream os;
    double rate = 1. - static_cast<double>(mct()) / static_cast<double>(nct());
    os << "N : " << nc() << "\nM : " << mc() << "\nK : " << kc() << "\nNNZ : " << nnz() << "\n";
    os << "puncture[" << puncture.size() << "] : ";
    print_list(os, puncture);
    os << "\nshorten[" << shorten.size() << "] : ";
    print_list(os, shorten);
    os << "\nRate : " << rate << "\nN (transmitted) : " << nct() << "\nM (transmitted) : " << mct()
       << "\nK (transmitted) : " << kct() << "\n";
    return os.str();
}

} // namespace ldpc_amd
