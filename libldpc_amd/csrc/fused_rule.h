/*
 * fused_rule.h — which codes the fused form of the likelihood-ratio iteration (detmath.h, "Fused form") applies to, and the
 * order in which such a code's check nodes take their inputs.  Plain C, no dependencies: the plan builder (plan.cpp) and the
 * CPU oracle (oracle/ldpc_oracle.c, test infrastructure) both include it, so the rule — a property of the parity-check
 * matrix alone, never of a batch, a device or a launch — is stated once.
 *
 * H in CSR form: row i has neighbours rcol[rptr[i] .. rptr[i+1]) in file order; column j has cptr[j+1] - cptr[j] edges.
 */
#ifndef LDPC_AMD_FUSED_RULE_H
#define LDPC_AMD_FUSED_RULE_H

#define DM_FUSED_MAX_SLOTS 8191     /* message slots (edges not ending in a leaf): byte offsets fit 16 bits */
#define DM_FUSED_MAX_VN_BLOCKS 32   /* blocks of <= 64 variable nodes of equal degree >= 2 (eight per wave) */
#define DM_FUSED_MAX_LEAF_BLOCKS 16 /* blocks of <= 64 check nodes of one class that have a leaf (four per wave) */
#define DM_FUSED_MAX_VN_DEGREE 64

/* Canonical input order of one check node (detmath.h, "Fused form", item 1): neighbours of degree >= 3, then those of
   degree 2, then the leaf.  order[k] = position in the row's file order of the node's k-th input; *flip = bit k set where
   input k is a neighbour of degree 2; *leaf = 1 when the last input is a leaf.  Returns 0 for a row the form does not take
   (degree outside 2..4, more than one leaf, a leaf on a node of degree 2, an isolated column cannot occur in a row). */
static inline int dm_fused_row_order(int deg, const int *row_cols, const int *cptr, int *order, unsigned *flip, int *leaf)
{
    int k = 0, nleaf = 0, pass, j;
    if (deg < 2 || deg > 4)
        return 0;
    *flip = 0;
    for (pass = 0; pass < 3; ++pass)
        for (j = 0; j < deg; ++j)
        {
            const int d = cptr[row_cols[j] + 1] - cptr[row_cols[j]];
            const int cls = d >= 3 ? 0 : (d == 2 ? 1 : 2);
            if (cls != pass)
                continue;
            if (cls == 1)
                *flip |= 1u << k;
            if (cls == 2)
                ++nleaf;
            order[k++] = j;
        }
    *leaf = nleaf;
    return nleaf <= 1 && !(nleaf == 1 && deg < 3);
}

/* class of a check node: degree | leaf << 3 | flip << 4 (flip bits are contiguous: positions n3 .. n3 + n2 - 1) */
static inline unsigned dm_fused_class(int deg, unsigned flip, int leaf) { return (unsigned)deg | ((unsigned)leaf << 3) | (flip << 4); }
#define DM_FUSED_NUM_CLASS_KEYS 256

static inline int dm_fused_applies(int rows, int cols, const int *rptr, const int *rcol, const int *cptr)
{
    int class_count[DM_FUSED_NUM_CLASS_KEYS] = {0};
    int deg_count[DM_FUSED_MAX_VN_DEGREE + 1] = {0};
    int i, nleaf = 0, nnz, blocks;
    if (rows <= 0 || cols <= 0)
        return 0;
    nnz = rptr[rows];
    for (i = 0; i < cols; ++i)
    {
        const int d = cptr[i + 1] - cptr[i];
        if (d < 1 || d > DM_FUSED_MAX_VN_DEGREE)
            return 0; /* an isolated column, or wider than the kernels' rolled variable-node loop was sized for */
        ++deg_count[d];
        nleaf += d == 1;
    }
    if (nnz - nleaf > DM_FUSED_MAX_SLOTS || nnz - nleaf < 1)
        return 0;
    blocks = 0;
    for (i = 2; i <= DM_FUSED_MAX_VN_DEGREE; ++i)
        blocks += (deg_count[i] + 63) / 64;
    if (blocks > DM_FUSED_MAX_VN_BLOCKS || blocks < 1)
        return 0;
    for (i = 0; i < rows; ++i)
    {
        int order[4], leaf;
        unsigned flip;
        if (!dm_fused_row_order(rptr[i + 1] - rptr[i], rcol + rptr[i], cptr, order, &flip, &leaf))
            return 0;
        ++class_count[dm_fused_class(rptr[i + 1] - rptr[i], flip, leaf)];
    }
    blocks = 0;
    for (i = 0; i < DM_FUSED_NUM_CLASS_KEYS; ++i)
        if (i & 8)
            blocks += (class_count[i] + 63) / 64;
    return blocks <= DM_FUSED_MAX_LEAF_BLOCKS;
}

#endif /* LDPC_AMD_FUSED_RULE_H */
