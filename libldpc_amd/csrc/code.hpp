// code.hpp — LDPC code container: parity-check / generator matrices in file-order adjacency form.
//
// Host-side support for the MI355X BP decoder.  Mirrors the observable behaviour of the
// reference's ldpc_code + sparse_csr (src/core/ldpc.{h,cpp}, src/core/sparse.h) — file format,
// derived sizes (nc, mc, kc, nct, mct, kct), puncture/shorten lists, bit_pos, max_degree — but is
// laid out as flat CSR/CSC index arrays that the device plan (plan.hpp) is built from.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace ldpc_amd
{

// Sparse GF(2) matrix.  Edge e = (row[e], col[e]) in file order; rows/cols keep file order inside
// each neighbour list (reference: sparse.h:122-152).
struct SparseGF2
{
    int rows = 0, cols = 0;
    std::vector<int> erow, ecol;            // [nnz]
    std::vector<int> rptr, rcol, redge;     // CSR: per row neighbour column + edge index
    std::vector<int> cptr, crow, cedge;     // CSC: per column neighbour row + edge index
    int nnz() const { return static_cast<int>(erow.size()); }
    bool empty() const { return rows == 0 && cols == 0; }

    // throws std::runtime_error("can not open file for reading") like the reference
    void read(const std::string &path, int skip_lines);
    void build_adjacency();

    // result[i] ^= right[col] over row i (syndrome, sparse.h:201-211)
    void multiply_right(const uint8_t *right, uint8_t *result) const;
    // result[j] ^= left[row] over column j (encoding, sparse.h:163-172); result is NOT cleared
    void multiply_left(const uint8_t *left, uint8_t *result) const;
    int rank() const; // over GF(2)
};

struct LdpcCode
{
    SparseGF2 H, G;
    std::vector<int> puncture, shorten, bit_pos;
    int max_degree = 0;

    // ldpc.cpp:7-38: H is mandatory, G optional ("" = none).  Throws std::runtime_error.
    LdpcCode(const std::string &pc_file, const std::string &gen_file);

    int nc() const { return H.cols; }
    int mc() const { return H.rows; }
    int kc() const { return H.cols - H.rows; }
    int nnz() const { return H.nnz(); }
    int nct() const { return nc() - static_cast<int>(puncture.size()) - static_cast<int>(shorten.size()); }
    int mct() const { return mc() - static_cast<int>(puncture.size()); }
    int kct() const { return nct() - mct(); }
    bool has_G() const { return !G.empty(); }
    int min_cn_degree() const;
    int max_cn_degree() const;
    int max_vn_degree() const;
    std::string describe() const; // same text block the reference CLI prints (ldpc.cpp:111-130)
};

} // namespace ldpc_amd
