/*
 * ldpc_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY, see ldpc_oracle.h).
 *
 * Plain-C restatement of the reference hot path.  Every function cites the reference
 * file:line it follows (paths relative to heat1q/libldpc).  Third-party arithmetic the
 * reference pulls in from outside its tree is restated from its published behaviour:
 *   - libstdc++ 11.4 <random> (bits/random.h, bits/random.tcc): mt19937_64,
 *     generate_canonical<double,53>, normal_distribution (Marsaglia polar),
 *     bernoulli_distribution;
 *   - glibc 2.35 libm exp/log/sqrt/pow (used directly in ORC_MATH_LIBM mode).
 * Build with -ffp-contract=off (the reference is built for baseline x86-64, no FMA).
 */
#define _GNU_SOURCE
#include "ldpc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../libldpc_amd/csrc/detmath.h"
#include "../libldpc_amd/csrc/fused_rule.h"

/* ------------------------------------------------------------------------------------ */
/* sparse matrix with file-order adjacency (sparse.h:8-90)                               */
/* ------------------------------------------------------------------------------------ */
typedef struct
{
    int rows, cols, nnz;
    int *erow, *ecol;   /* edge list in file order                        */
    int *rptr, *rnode, *redge; /* per row: neighbour column + edge index, file order */
    int *cptr, *cnode, *cedge; /* per col: neighbour row + edge index, file order    */
} spm;

struct orc_code
{
    spm H, G;
    int hasG;
    int npunct, nshort;
    int *punct, *shrt;
    int *bit_pos;
    int nct, mct, kct, kc, max_degree;
    int n_bitpos; /* entries of bit_pos: > nct when a column is both punctured and shortened */
};

static void spm_free(spm *m)
{
    free(m->erow), free(m->ecol);
    free(m->rptr), free(m->rnode), free(m->redge);
    free(m->cptr), free(m->cnode), free(m->cedge);
    memset(m, 0, sizeof *m);
}

static int line_is_blank(const char *s)
{
    for (; *s; ++s)
        if (*s != ' ' && *s != '\t' && *s != '\r' && *s != '\n')
            return 0;
    return 1;
}

/* sparse.h:92-153: body lines "row col [value]", dims = max index + 1, edge index = line order */
static int spm_read(spm *m, const char *path, int skip_lines)
{
    FILE *fp = fopen(path, "r");
    if (!fp)
        return -1;
    char *line = NULL;
    size_t cap = 0;
    int n = 0, capn = 0;
    int *er = NULL, *ec = NULL;
    int maxr = 0, maxc = 0;
    while (skip_lines-- > 0)
        if (getline(&line, &cap, fp) < 0)
            break;
    while (getline(&line, &cap, fp) >= 0)
    {
        /* a blank line is undefined behaviour in the reference parser (uninitialised
           edge); both this oracle and the product skip it */
        if (line_is_blank(line))
            continue;
        int r = 0, c = 0;
        sscanf(line, "%d %d", &r, &c);
        if (n == capn)
        {
            capn = capn ? capn * 2 : 4096;
            er = realloc(er, sizeof(int) * capn);
            ec = realloc(ec, sizeof(int) * capn);
        }
        er[n] = r, ec[n] = c;
        if (r > maxr)
            maxr = r;
        if (c > maxc)
            maxc = c;
        ++n;
    }
    free(line);
    fclose(fp);
    m->rows = maxr + 1;
    m->cols = maxc + 1;
    m->nnz = n;
    m->erow = er, m->ecol = ec;
    m->rptr = calloc(m->rows + 1, sizeof(int));
    m->cptr = calloc(m->cols + 1, sizeof(int));
    for (int e = 0; e < n; ++e)
        m->rptr[er[e] + 1]++, m->cptr[ec[e] + 1]++;
    for (int i = 0; i < m->rows; ++i)
        m->rptr[i + 1] += m->rptr[i];
    for (int j = 0; j < m->cols; ++j)
        m->cptr[j + 1] += m->cptr[j];
    m->rnode = malloc(sizeof(int) * (n ? n : 1));
    m->redge = malloc(sizeof(int) * (n ? n : 1));
    m->cnode = malloc(sizeof(int) * (n ? n : 1));
    m->cedge = malloc(sizeof(int) * (n ? n : 1));
    int *rf = calloc(m->rows, sizeof(int)), *cf = calloc(m->cols, sizeof(int));
    for (int e = 0; e < n; ++e) /* stable: neighbour lists keep file order */
    {
        int r = er[e], c = ec[e];
        int pr = m->rptr[r] + rf[r]++, pc = m->cptr[c] + cf[c]++;
        m->rnode[pr] = c, m->redge[pr] = e;
        m->cnode[pc] = r, m->cedge[pc] = e;
    }
    free(rf), free(cf);
    return 0;
}

static int contains(const int *a, int n, int v)
{
    for (int i = 0; i < n; ++i)
        if (a[i] == v)
            return 1;
    return 0;
}

/* ldpc.cpp:40-101 */
orc_code *orc_code_load(const char *pc_file, const char *gen_file)
{
    FILE *fp = fopen(pc_file, "r");
    if (!fp)
        return NULL;
    orc_code *c = calloc(1, sizeof *c);
    char *line = NULL;
    size_t cap = 0;
    int skip = 0;
    while (getline(&line, &cap, fp) >= 0)
    {
        char *colon = strchr(line, ':');
        if (!colon)
            break;
        *colon = 0;
        int is_p = strstr(line, "puncture") != NULL;
        int is_s = !is_p && strstr(line, "shorten") != NULL;
        if (is_p || is_s)
        {
            char *p = colon + 1, *end;
            for (;;)
            {
                long v = strtol(p, &end, 10);
                if (end == p)
                    break;
                p = end;
                if (is_p)
                {
                    c->punct = realloc(c->punct, sizeof(int) * (c->npunct + 1));
                    c->punct[c->npunct++] = (int)v;
                }
                else
                {
                    c->shrt = realloc(c->shrt, sizeof(int) * (c->nshort + 1));
                    c->shrt[c->nshort++] = (int)v;
                }
            }
        }
        ++skip;
    }
    free(line);
    fclose(fp);
    if (spm_read(&c->H, pc_file, skip) != 0)
    {
        orc_code_free(c);
        return NULL;
    }
    int md = 0;
    for (int i = 0; i < c->H.rows; ++i)
        if (c->H.rptr[i + 1] - c->H.rptr[i] > md)
            md = c->H.rptr[i + 1] - c->H.rptr[i];
    for (int j = 0; j < c->H.cols; ++j)
        if (c->H.cptr[j + 1] - c->H.cptr[j] > md)
            md = c->H.cptr[j + 1] - c->H.cptr[j];
    c->max_degree = md;
    c->bit_pos = malloc(sizeof(int) * c->H.cols);
    int nb = 0;
    for (int i = 0; i < c->H.cols; ++i)
    {
        if (contains(c->shrt, c->nshort, i) || contains(c->punct, c->npunct, i))
            continue;
        c->bit_pos[nb++] = i;
    }
    c->n_bitpos = nb;
    /* ldpc.h:47-59 (note: nct is computed from the list sizes, not from bit_pos) */
    c->kc = c->H.cols - c->H.rows;
    c->nct = c->H.cols - c->npunct - c->nshort;
    c->mct = c->H.rows - c->npunct;
    c->kct = c->nct - c->mct;
    if (gen_file && gen_file[0])
    {
        if (spm_read(&c->G, gen_file, 0) != 0)
        {
            orc_code_free(c);
            return NULL;
        }
        c->hasG = 1;
    }
    return c;
}

void orc_code_free(orc_code *c)
{
    if (!c)
        return;
    spm_free(&c->H);
    spm_free(&c->G);
    free(c->punct), free(c->shrt), free(c->bit_pos);
    free(c);
}

int orc_code_nc(const orc_code *c) { return c->H.cols; }
int orc_code_mc(const orc_code *c) { return c->H.rows; }
int orc_code_kc(const orc_code *c) { return c->kc; }
int orc_code_nnz(const orc_code *c) { return c->H.nnz; }
int orc_code_nct(const orc_code *c) { return c->nct; }
int orc_code_mct(const orc_code *c) { return c->mct; }
int orc_code_kct(const orc_code *c) { return c->kct; }
int orc_code_max_degree(const orc_code *c) { return c->max_degree; }
int orc_code_num_puncture(const orc_code *c) { return c->npunct; }
int orc_code_num_shorten(const orc_code *c) { return c->nshort; }
int orc_code_has_G(const orc_code *c) { return c->hasG; }
int orc_code_g_rows(const orc_code *c) { return c->hasG ? c->G.rows : 0; }
int orc_code_g_cols(const orc_code *c) { return c->hasG ? c->G.cols : 0; }
int orc_code_g_nnz(const orc_code *c) { return c->hasG ? c->G.nnz : 0; }
void orc_code_edges(const orc_code *c, int *er, int *ec)
{
    memcpy(er, c->H.erow, sizeof(int) * c->H.nnz);
    memcpy(ec, c->H.ecol, sizeof(int) * c->H.nnz);
}
void orc_code_bit_pos(const orc_code *c, int *b) { memcpy(b, c->bit_pos, sizeof(int) * c->nct); }
int orc_code_n_bitpos(const orc_code *c) { return c->n_bitpos; }
void orc_code_puncture(const orc_code *c, int *p) { memcpy(p, c->punct, sizeof(int) * c->npunct); }
void orc_code_shorten(const orc_code *c, int *s) { memcpy(s, c->shrt, sizeof(int) * c->nshort); }

/* sparse.h:201-211: result[i] += right[n] * value over row neighbours (GF(2), result starts 0) */
void orc_syndrome(const orc_code *c, const uint8_t *word, uint8_t *synd)
{
    const spm *H = &c->H;
    for (int i = 0; i < H->rows; ++i)
    {
        uint8_t s = 0;
        for (int p = H->rptr[i]; p < H->rptr[i + 1]; ++p)
            s ^= (uint8_t)(word[H->rnode[p]] != 0);
        synd[i] = s;
    }
}

/* sparse.h:163-172: cw[j] += info[row] over column neighbours of G; cw is NOT cleared */
void orc_encode_accumulate(const orc_code *c, const uint8_t *info, uint8_t *cw)
{
    const spm *G = &c->G;
    for (int j = 0; j < G->cols; ++j)
        for (int p = G->cptr[j]; p < G->cptr[j + 1]; ++p)
            cw[j] ^= (uint8_t)(info[G->cnode[p]] != 0);
}

/* GF(2) rank by dense bitset elimination; the reference's sparse elimination
   (sparse.h:233-300) computes the same number */
int orc_rank(const orc_code *c)
{
    const spm *H = &c->H;
    int words = (H->cols + 63) / 64;
    uint64_t *a = calloc((size_t)H->rows * words, 8);
    for (int e = 0; e < H->nnz; ++e)
        a[(size_t)H->erow[e] * words + H->ecol[e] / 64] ^= 1ull << (H->ecol[e] % 64);
    int rank = 0;
    for (int col = 0; col < H->cols && rank < H->rows; ++col)
    {
        int piv = -1;
        for (int r = rank; r < H->rows; ++r)
            if (a[(size_t)r * words + col / 64] >> (col % 64) & 1)
            {
                piv = r;
                break;
            }
        if (piv < 0)
            continue;
        if (piv != rank)
            for (int w = 0; w < words; ++w)
            {
                uint64_t t = a[(size_t)piv * words + w];
                a[(size_t)piv * words + w] = a[(size_t)rank * words + w];
                a[(size_t)rank * words + w] = t;
            }
        for (int r = rank + 1; r < H->rows; ++r)
            if (a[(size_t)r * words + col / 64] >> (col % 64) & 1)
                for (int w = col / 64; w < words; ++w)
                    a[(size_t)r * words + w] ^= a[(size_t)rank * words + w];
        ++rank;
    }
    free(a);
    return rank;
}

/* ------------------------------------------------------------------------------------ */
/* math dispatch                                                                         */
/* ------------------------------------------------------------------------------------ */
double orc_exp(int mode, double x) { return mode == ORC_MATH_DET ? dm_exp(x) : exp(x); }
double orc_log(int mode, double x) { return mode == ORC_MATH_DET ? dm_log(x) : log(x); }

/* decoder.h:7-10 */
static inline int sgn(double x) { return 1 - 2 * (int)(signbit(x) != 0); }
/* std::min(a, b) = (b < a) ? b : a */
static inline double stdmin(double a, double b) { return (b < a) ? b : a; }

/* decoder.h:12-15 */
static double jacobian_libm(double x, double y)
{
    return sgn(x) * sgn(y) * stdmin(fabs(x), fabs(y)) +
           log((1 + exp(-fabs(x + y))) / (1 + exp(-fabs(x - y))));
}
/* same expression with the deterministic exp/log pair of detmath.h (dm_boxplus spells it out) */
static double jacobian_det(double x, double y) { return dm_boxplus(x, y); }
/* decoder.h:17-20 */
static double minsum(double x, double y) { return sgn(x) * sgn(y) * stdmin(fabs(x), fabs(y)); }

typedef double (*cn_fn)(double, double);

/* ------------------------------------------------------------------------------------ */
/* BP decoder state (decoder.h:26-105)                                                   */
/* ------------------------------------------------------------------------------------ */
typedef struct
{
    const orc_code *code;
    cn_fn cn;
    int early_term;
    unsigned iterations;
    double *v2c, *c2v, *F, *B, *llr_in, *llr_out;
    uint8_t *co;
} dec_t;

static void dec_init(dec_t *d, const orc_code *c, int min_sum, int early_term, unsigned iters, int math)
{
    d->code = c;
    d->cn = min_sum ? minsum : (math == ORC_MATH_DET ? jacobian_det : jacobian_libm);
    d->early_term = early_term;
    d->iterations = iters;
    int nnz = c->H.nnz, nc = c->H.cols, md = c->max_degree > 2 ? c->max_degree : 2;
    d->v2c = calloc(nnz, 8), d->c2v = calloc(nnz, 8);
    d->F = calloc(md, 8), d->B = calloc(md, 8);
    d->llr_in = calloc(nc, 8), d->llr_out = calloc(nc, 8);
    d->co = calloc(nc, 1);
}
static void dec_free(dec_t *d)
{
    free(d->v2c), free(d->c2v), free(d->F), free(d->B), free(d->llr_in), free(d->llr_out), free(d->co);
}

/*
 * ORC_MATH_DET only: the forward/backward recursion (decoder.cpp:31-44) carried in E = e^-|L| (detmath.h,
 * dm_e_combine / dm_e_to_llr) — the arithmetic the HIP kernel runs — or, when every input is large and they lie
 * close together, the saturated form (detmath.h, dm_sat_*).  Returns 0 (nothing done) when neither applies (an input
 * exceeds DM_SHARED_LIMIT); the caller then evaluates this node with dm_boxplus, as the kernel does.
 * F[cw-1] and B[0], which the reference computes and never reads, are not evaluated.
 */
static int cn_update_det_shared(dec_t *d, const int *cn, int cw)
{
    enum { MAXD = 64 };
    double v[MAXD], ev[MAXD];
    uint32_t sv[MAXD], sF[MAXD], sB[MAXD];
    if (cw > MAXD || cw < 3) /* a degree-2 node only swaps its inputs: generic path */
        return 0;
    double amax = 0.0, mu = HUGE_VAL;
    for (int j = 0; j < cw; ++j)
    {
        v[j] = d->v2c[cn[j]];
        amax = fmax(amax, fabs(v[j]));
        mu = fmin(mu, fabs(v[j]));
    }
    const int two_base = cw >= 5 && cw <= 16 && dm_sat2_applies(mu, amax); /* (detmath.h: degrees 5..16) */
    if (dm_sat_applies(mu, amax) || two_base)
    {
        /* saturated form (detmath.h): sums of e^-(|v| - mu); F[j] = inputs 0..j, B[j] = inputs j..cw-1 */
        double Fs[MAXD], Bs[MAXD];
        for (int j = 0; j < cw; ++j)
        {
            Fs[j] = dm_sat_e(fabs(v[j]), mu);
            sv[j] = DM_SIGN_WORD(v[j]);
        }
        sF[0] = sv[0], sB[cw - 1] = sv[cw - 1];
        Bs[cw - 1] = Fs[cw - 1];
        for (int j = cw - 2; j >= 1; --j)
            Bs[j] = Bs[j + 1] + Fs[j], sB[j] = sB[j + 1] ^ sv[j];
        for (int j = 1; j < cw - 1; ++j)
            Fs[j] = Fs[j - 1] + Fs[j], sF[j] = sF[j - 1] ^ sv[j];
        d->c2v[cn[0]] = dm_sat_llr(sB[1], mu, Bs[1]);
        d->c2v[cn[cw - 1]] = dm_sat_llr(sF[cw - 2], mu, Fs[cw - 2]);
        for (int j = 1; j < cw - 1; ++j)
            d->c2v[cn[j]] = dm_sat_llr(sF[j - 1] ^ sB[j + 1], mu, Fs[j - 1] + Bs[j + 1]);
        if (two_base)
        {
            /* the (first) edge that holds the minimum takes its output from base m2 (detmath.h, dm_sat2_applies) */
            int jm = 0;
            while (fabs(v[jm]) != mu)
                ++jm;
            double m2 = HUGE_VAL, own = 0.0;
            uint32_t sign_all = 0;
            for (int j = 0; j < cw; ++j)
                if (j != jm)
                    m2 = fmin(m2, fabs(v[j]));
            for (int j = 0; j < cw; ++j)
            {
                own += j == jm ? 0.0 : dm_sat_e(fabs(v[j]), m2);
                sign_all ^= sv[j];
            }
            d->c2v[cn[jm]] = dm_sat_llr(sign_all ^ sv[jm], m2, own);
        }
        return 1;
    }
    for (int j = 0; j < cw; ++j)
        if (!(fabs(v[j]) <= DM_SHARED_LIMIT))
            return 0;
    for (int j = 0; j < cw; ++j)
    {
        ev[j] = dm_boxplus_exp(fabs(v[j]));
        sv[j] = DM_SIGN_WORD(v[j]);
    }
    sF[0] = sv[0], sB[cw - 1] = sv[cw - 1];
    for (int j = 1; j < cw - 1; ++j)
        sF[j] = sF[j - 1] ^ sv[j];
    for (int j = cw - 2; j >= 1; --j)
        sB[j] = sB[j + 1] ^ sv[j];
    if (cw == 3)
    {
        d->c2v[cn[0]] = dm_e_to_llr(sB[1], dm_e_combine(ev[2], ev[1]));
        d->c2v[cn[2]] = dm_e_to_llr(sF[1], dm_e_combine(ev[0], ev[1]));
        d->c2v[cn[1]] = dm_e_to_llr(sF[0] ^ sB[2], dm_e_combine(ev[0], ev[2]));
        return 1;
    }
    /* cw >= 4: partial results as undivided fractions (detmath.h, dm_efrac); F[j] = inputs 0..j, B[j] = inputs j..cw-1 */
    dm_efrac Ff[MAXD], Bf[MAXD];
    Ff[1] = dm_efrac_first(ev[0], ev[1]);
    Bf[cw - 2] = dm_efrac_first(ev[cw - 1], ev[cw - 2]);
    for (int j = 2; j <= cw - 2; ++j)
        Ff[j] = dm_efrac_step(Ff[j - 1], ev[j]);
    for (int j = cw - 3; j >= 1; --j)
        Bf[j] = dm_efrac_step(Bf[j + 1], ev[j]);
    d->c2v[cn[0]] = dm_e_to_llr(sB[1], dm_efrac_e(Bf[1]));
    d->c2v[cn[cw - 1]] = dm_e_to_llr(sF[cw - 2], dm_efrac_e(Ff[cw - 2]));
    d->c2v[cn[1]] = dm_e_to_llr(sF[0] ^ sB[2], dm_efrac_e(dm_efrac_step(Bf[2], ev[0])));
    d->c2v[cn[cw - 2]] = dm_e_to_llr(sF[cw - 3] ^ sB[cw - 1], dm_efrac_e(dm_efrac_step(Ff[cw - 3], ev[cw - 1])));
    for (int j = 2; j <= cw - 3; ++j)
        d->c2v[cn[j]] = dm_e_to_llr(sF[j - 1] ^ sB[j + 1], dm_efrac_e2(Ff[j - 1], Bf[j + 1]));
    return 1;
}

/* decoder.h:47-64 */
static int is_codeword(const dec_t *d)
{
    const spm *H = &d->code->H;
    for (int i = 0; i < H->rows; ++i)
    {
        uint8_t s = 0;
        for (int p = H->rptr[i]; p < H->rptr[i + 1]; ++p)
            s ^= d->co[H->rnode[p]];
        if (s != 0)
            return 0;
    }
    return 1;
}

/*
 * ORC_MATH_DET only: the sum-product iteration in likelihood-ratio form (detmath.h, "Likelihood-ratio form") —
 * the arithmetic the HIP kernels run for BP with early termination.  v2c[] holds rho = e^L, c2v[] holds
 * lambda = e^-L.  Same schedule, same orders of accumulation as decoder.cpp:11-78.  Returns -1 when a value
 * of the frame leaves the representable box; the caller then decodes the frame with the LLR-domain form.
 */
/* shared: nodes of degree 3 and 4 take one reciprocal of the product of their denominators (detmath.h, "Shared-reciprocal
   check nodes": what the kernels run WITH early termination); returns nonzero when such a product left its range */
/* shared6: nodes of degree 6 take two reciprocals for their six outputs (dm_cn6_shared: what the kernels run WITH early
   termination for codes their LDS-resident decoder does not take); *esc6 is set when such a product left its range — the
   caller lets that count only once the frame has gone on to the variable-node pass */
static int cn_update_ratio(dec_t *d, const int *cn, int cw, int shared, int shared6, int *esc6)
{
    enum { MAXD = 64 };
    double v[MAXD] = {0};
    for (int j = 0; j < cw; ++j)
        v[j] = d->v2c[cn[j]];
    if (shared6 && cw == 6)
    {
        const uint32_t p_hi = dm_cn6_shared(v);
        for (int j = 0; j < cw; ++j)
            d->c2v[cn[j]] = v[j];
        *esc6 |= DM_SHARED_OVERFLOW(p_hi);
        return 0;
    }
    if (shared && (cw == 3 || cw == 4))
    {
        const uint32_t p_hi = cw == 3 ? dm_cn3_shared(v) : dm_cn4_shared(v);
        for (int j = 0; j < cw; ++j)
            d->c2v[cn[j]] = v[j];
        return DM_SHARED_OVERFLOW(p_hi);
    }
    if (cw == 2)
    {
        d->c2v[cn[0]] = 1.0 / v[1];
        d->c2v[cn[1]] = 1.0 / v[0];
    }
    else if (cw == 3)
    {
        d->c2v[cn[0]] = dm_ratio_lambda(v[2], v[1]);
        d->c2v[cn[1]] = dm_ratio_lambda(v[0], v[2]);
        d->c2v[cn[2]] = dm_ratio_lambda(v[0], v[1]);
    }
    else if (cw == 4)
    {
        double nF = DM_FMA(v[0], v[1], 1.0), dF = v[0] + v[1];
        double nB = DM_FMA(v[3], v[2], 1.0), dB = v[3] + v[2];
        d->c2v[cn[0]] = dm_ratio_lambda_frac(nB, dB, v[1]);
        d->c2v[cn[1]] = dm_ratio_lambda_frac(nB, dB, v[0]);
        d->c2v[cn[2]] = dm_ratio_lambda_frac(nF, dF, v[3]);
        d->c2v[cn[3]] = dm_ratio_lambda_frac(nF, dF, v[2]);
    }
    else
    {
        /* cw >= 5: partial results as undivided fractions (detmath.h, dm_frac), F[j] = inputs 0..j,
           B[j] = inputs j..cw-1, rescaled when they cover an odd number >= 3 of inputs */
        dm_frac Ff[MAXD], Bf[MAXD];
        Ff[1] = dm_frac_first(v[0], v[1]);
        Bf[cw - 2] = dm_frac_first(v[cw - 1], v[cw - 2]);
        for (int j = 2; j <= cw - 3; ++j)
        {
            Ff[j] = dm_frac_step(Ff[j - 1], v[j]);
            if ((j + 1) % 2 == 1)
                Ff[j] = dm_frac_norm(Ff[j]);
        }
        for (int j = cw - 3; j >= 2; --j)
        {
            Bf[j] = dm_frac_step(Bf[j + 1], v[j]);
            if ((cw - j) % 2 == 1)
                Bf[j] = dm_frac_norm(Bf[j]);
        }
        d->c2v[cn[0]] = dm_ratio_lambda_frac(Bf[2].n, Bf[2].d, v[1]);
        d->c2v[cn[1]] = dm_ratio_lambda_frac(Bf[2].n, Bf[2].d, v[0]);
        d->c2v[cn[cw - 2]] = dm_ratio_lambda_frac(Ff[cw - 3].n, Ff[cw - 3].d, v[cw - 1]);
        d->c2v[cn[cw - 1]] = dm_ratio_lambda_frac(Ff[cw - 3].n, Ff[cw - 3].d, v[cw - 2]);
        for (int j = 2; j <= cw - 3; ++j)
            d->c2v[cn[j]] = dm_frac_lambda2(Ff[j - 1], Bf[j + 1]);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Element-by-element arithmetic for tests/test_gpu_math.py (same function numbering as MathFn in
 * libldpc_amd/csrc/kernels.hpp).
 *   orc_math_ref: what the function MEANS, computed without detmath.h — glibc libm in binary64 for exp / log / the
 *                 reference's jacobian expression (decoder.h:12-15), x87 extended precision (long double) for the
 *                 rational expressions, the plain divided forward/backward recursion of decoder.cpp:31-44 in long
 *                 double for the check-node updates.
 *   orc_math_det: detmath.h compiled for the host (the det-mode oracle's arithmetic).
 * ------------------------------------------------------------------------------------------------ */
static double ref_jacobian(double x, double y) /* decoder.h:12-15, libm */
{
    int sx = 1 - 2 * (signbit(x) != 0), sy = 1 - 2 * (signbit(y) != 0);
    double ax = fabs(x), ay = fabs(y);
    double mn = ay < ax ? ay : ax;
    return (double)(sx * sy) * mn + log((1 + exp(-fabs(x + y))) / (1 + exp(-fabs(x - y))));
}

static void ref_cn_ratio_ld(int cw, const double *v, double *out) /* rho in, lambda out; divided at every step */
{
    long double F[16], B[16];
    F[0] = v[0], B[cw - 1] = v[cw - 1];
    for (int j = 1; j < cw; ++j)
    {
        F[j] = (1.0L + F[j - 1] * (long double)v[j]) / (F[j - 1] + (long double)v[j]);
        B[cw - 1 - j] = (1.0L + B[cw - j] * (long double)v[cw - 1 - j]) / (B[cw - j] + (long double)v[cw - 1 - j]);
    }
    out[0] = (double)(1.0L / B[1]);
    out[cw - 1] = (double)(1.0L / F[cw - 2]);
    for (int j = 1; j < cw - 1; ++j)
        out[j] = (double)((F[j - 1] + B[j + 1]) / (1.0L + F[j - 1] * B[j + 1]));
}

static void ref_cn_llr(int cw, const double *v, double *out) /* decoder.cpp:31-44 with the libm jacobian */
{
    double F[16], B[16];
    F[0] = v[0], B[cw - 1] = v[cw - 1];
    for (int j = 1; j < cw; ++j)
    {
        F[j] = ref_jacobian(F[j - 1], v[j]);
        B[cw - 1 - j] = ref_jacobian(B[cw - j], v[cw - j - 1]);
    }
    out[0] = B[1], out[cw - 1] = F[cw - 2];
    for (int j = 1; j < cw - 1; ++j)
        out[j] = ref_jacobian(F[j - 1], B[j + 1]);
}

static int math_width(int fn)
{
    static const int w[] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 3, 4, 5, 6, 8, 4, 6, 3, 4, 6};
    return fn >= 0 && fn < (int)(sizeof w / sizeof w[0]) ? w[fn] : 0;
}

int orc_math_ref(int fn, uint64_t n, const double *a, const double *b, double *out)
{
    const int w = math_width(fn);
    if (!w)
        return -1;
    for (uint64_t i = 0; i < n; ++i)
    {
        long double x = a[i * w], y = b ? b[i * w] : 0;
        switch (fn)
        {
        case 0: out[i] = exp(a[i]); break;
        case 1: out[i] = log(a[i]); break;
        case 2: out[i] = ref_jacobian(a[i], b[i]); break;
        case 3: out[i] = a[i] / b[i]; break;
        case 4: out[i] = (double)((1.0L + x * y) / (x + y)); break;
        case 5: out[i] = (double)((x + y) / (1.0L + x * y)); break;
        case 6: out[i] = (double)((x + y) / (1.0L + x * y)); break;
        case 7: out[i] = exp(fmin(fmax(a[i], -700.0), 700.0)); break;
        case 8: out[i] = exp(-fmin(a[i], 700.0)); break;
        case 9: out[i] = log(a[i]); break;
        case 15: case 16: ref_cn_llr(w, a + i * w, out + i * w); break;
        default: ref_cn_ratio_ld(w, a + i * w, out + i * w); break;
        }
    }
    return 0;
}

static void det_cn_ratio(int cw, const double *v, double *out)
{
    dec_t d;
    double v2c[16], c2v[16];
    int cn[16];
    for (int j = 0; j < cw; ++j)
        v2c[j] = v[j], cn[j] = j;
    d.v2c = v2c, d.c2v = c2v;
    int e6 = 0;
    cn_update_ratio(&d, cn, cw, 0, 0, &e6);
    for (int j = 0; j < cw; ++j)
        out[j] = c2v[j];
}

static void det_cn_shared(int cw, const double *v, double *out) /* rows whose denominator product overflows: NaN */
{
    dec_t d;
    double v2c[16], c2v[16];
    int cn[16];
    for (int j = 0; j < cw; ++j)
        v2c[j] = v[j], cn[j] = j;
    d.v2c = v2c, d.c2v = c2v;
    int e6 = 0;
    const int over = cn_update_ratio(&d, cn, cw, cw != 6, cw == 6, &e6);
    for (int j = 0; j < cw; ++j)
        out[j] = (over || e6) ? NAN : c2v[j];
}

int orc_math_det(int fn, uint64_t n, const double *a, const double *b, double *out)
{
    const int w = math_width(fn);
    if (!w)
        return -1;
    for (uint64_t i = 0; i < n; ++i)
        switch (fn)
        {
        case 0: out[i] = dm_exp(a[i]); break;
        case 1: out[i] = dm_log(a[i]); break;
        case 2: out[i] = dm_boxplus(a[i], b[i]); break;
        case 3: out[i] = dm_ratio_div(a[i], b[i]); break;
        case 4: out[i] = dm_ratio_rho(a[i], b[i]); break;
        case 5: out[i] = dm_ratio_lambda(a[i], b[i]); break;
        case 6: out[i] = dm_e_combine(a[i], b[i]); break;
        case 7: out[i] = dm_exp_clamped(a[i]); break;
        case 8: out[i] = dm_boxplus_exp(a[i]); break;
        case 9: out[i] = dm_boxplus_log(a[i]); break;
        case 15: case 16:
        {
            dec_t d;
            double v2c[16], c2v[16], F[16], B[16];
            int cn[16];
            for (int j = 0; j < w; ++j)
                v2c[j] = a[i * w + j], cn[j] = j;
            d.v2c = v2c, d.c2v = c2v, d.F = F, d.B = B, d.cn = jacobian_det;
            if (!cn_update_det_shared(&d, cn, w))
            {
                F[0] = v2c[0], B[w - 1] = v2c[w - 1];
                for (int j = 1; j < w; ++j)
                {
                    F[j] = jacobian_det(F[j - 1], v2c[j]);
                    B[w - 1 - j] = jacobian_det(B[w - j], v2c[w - j - 1]);
                }
                c2v[0] = B[1], c2v[w - 1] = F[w - 2];
                for (int j = 1; j < w - 1; ++j)
                    c2v[j] = jacobian_det(F[j - 1], B[j + 1]);
            }
            for (int j = 0; j < w; ++j)
                out[i * w + j] = c2v[j];
            break;
        }
        case 17: case 18: case 19: det_cn_shared(w, a + i * w, out + i * w); break;
        default: det_cn_ratio(w, a + i * w, out + i * w); break;
        }
    return 0;
}

/* LLR-domain form from iteration `first` on; skip_first_cn: the c2v messages of iteration `first` are already in place
   (hand-over from the ratio form, detmath.h "Hand-over") */
static int dec_decode_llr_from(dec_t *d, unsigned first, int skip_first_cn);

/* The hand-over form is what the kernels run WITHOUT early termination for codes their LDS-resident decoder takes
   (libldpc_amd/csrc/plan.cpp: check nodes up to degree 8 and one frame's binary64 messages, input LLRs and hard bits
   within 160 KB); larger codes run the LLR-domain form for all iterations then.  A property of the code alone. */
static int handover_applies(const orc_code *c)
{
    const spm *H = &c->H;
    int max_cw = 0;
    for (int i = 0; i < H->rows; ++i)
        if (H->rptr[i + 1] - H->rptr[i] > max_cw)
            max_cw = H->rptr[i + 1] - H->rptr[i];
    const size_t lds = (size_t)8 * H->nnz + (size_t)8 * H->cols + (size_t)((H->nnz + 15) / 16) * 16 + 16;
    return max_cw <= 8 && lds <= (size_t)160 * 1024;
}

/* returns the iteration count, or -1 when a value left the representable box (the caller decodes the frame again in the
   LLR domain).  early_term off: the frame may be handed over to the LLR-domain form mid-way (detmath.h "Hand-over"). */
static int dec_decode_ratio(dec_t *d, int allow_shared)
{
    const spm *H = &d->code->H;
    double *lam = malloc(8 * (size_t)(H->cols > 0 ? H->cols : 1));
    int escaped = 0;
    int32_t ho_key = 0;
    for (int i = 0; i < H->cols; ++i)
    {
        if (H->cptr[i + 1] == H->cptr[i])
            continue; /* isolated variable node: its LLR multiplies nothing */
        double L = d->llr_in[i];
        escaped |= !(fabs(L) <= DM_RATIO_LLR_LIMIT);
        lam[i] = dm_exp_clamped(0.0 - L);
        double v0 = 1.0 / lam[i];
        for (int p = H->cptr[i]; p < H->cptr[i + 1]; ++p)
            d->v2c[H->cedge[p]] = v0;
    }
    /* shared-reciprocal check nodes (detmath.h): with early termination; degrees 3 and 4 for codes the LDS-resident decoder
       takes, degree 6 for the others */
    const int shared = allow_shared && d->early_term && handover_applies(d->code);
    const int shared6 = allow_shared && d->early_term && !handover_applies(d->code);
    unsigned I = 0;
    int ret = -1;
    for (;;)
    {
        /* loop pass I: CN pass I, then the checks on what VN pass I-1 left behind, then VN pass I */
        int esc6 = 0;
        for (int i = 0; i < H->rows && !escaped; ++i)
        {
            int cw = H->rptr[i + 1] - H->rptr[i];
            if (cw > 16 || cw < 2) /* nodes wider than the kernels' register tiles: LLR-domain form only */
            {
                escaped = 1;
                break;
            }
            escaped |= cn_update_ratio(d, H->redge + H->rptr[i], cw, shared, shared6, &esc6);
        }
        if (escaped)
            break;
        if (I > 0 && d->early_term && is_codeword(d))
        {
            ret = (int)I - 1;
            break;
        }
        if (I == d->iterations)
        {
            ret = (int)I;
            break;
        }
        if (!d->early_term && DM_HANDOVER_DUE(ho_key))
        {
            for (int e = 0; e < H->nnz; ++e)
                d->c2v[e] = 0.0 - dm_log(d->c2v[e]);
            free(lam);
            return dec_decode_llr_from(d, I, 1);
        }
        if (esc6) /* a degree-6 node's product left its range in a pass whose outputs the frame goes on to use */
        {
            escaped = 1;
            break;
        }
        for (int i = 0; i < H->cols && !escaped; ++i)
        {
            int deg = H->cptr[i + 1] - H->cptr[i];
            if (deg == 0)
            {
                d->llr_out[i] = d->llr_in[i];
                d->co[i] = (uint8_t)(d->llr_in[i] <= 0);
                continue;
            }
            if (deg == 1)
            {
                /* a leaf: out - c2v = L_ch, so its v2c message is the channel ratio itself (no division, nothing to
                   range-check); the decision total <= 0 is lambda_ch * lambda(c2v) >= 1, taken as lambda(c2v) >= rho_ch */
                int e = H->cedge[H->cptr[i]];
                double c = d->c2v[e], rho = 1.0 / lam[i];
                d->co[i] = (uint8_t)(c >= rho);
                d->llr_out[i] = 0.0 - dm_log(c / rho);
                d->v2c[e] = rho;
                continue;
            }
            double prod = lam[i];
            for (int k = 0; k < deg; ++k)
            {
                prod *= d->c2v[H->cedge[H->cptr[i] + k]];
                if (deg > 3 && k % 3 == 2)
                    escaped |= dm_ratio_out_of_range(prod);
            }
            int32_t hk = dm_handover_key(prod);
            ho_key = hk > ho_key ? hk : ho_key;
            d->co[i] = (uint8_t)(prod >= 1.0);
            d->llr_out[i] = 0.0 - dm_log(prod);
            double tot = 1.0 / prod;
            for (int k = 0; k < deg; ++k)
            {
                int e = H->cedge[H->cptr[i] + k];
                double o = tot * d->c2v[e];
                escaped |= dm_ratio_out_of_range(o);
                d->v2c[e] = o;
            }
        }
        ++I;
    }
    free(lam);
    return ret;
}

/* The fused form (detmath.h "Fused form", fused_rule.h): what the kernels' FIRST launch runs for sum-product with early
   termination on codes the rule takes — check nodes take their inputs in the rule's order, a leaf's decision is taken by its
   check node (n >= rho_ch d), degree-2 variable nodes receive rho(c2v) and multiply.  Same schedule as dec_decode_ratio;
   v2c[] holds rho(v2c) (a leaf's edge: its constant rho_ch), c2v[] holds lambda(c2v), or rho(c2v) on edges into variable
   nodes of degree 2 (nothing on a leaf's edge).  Returns the iteration count, or -1 when a value left its range. */
static int fused_applies(const orc_code *c)
{
    const spm *H = &c->H;
    return handover_applies(c) && dm_fused_applies(H->rows, H->cols, H->rptr, H->rnode, H->cptr);
}

/* one check-node pass of the fused form; handover: the pass that ends the ratio form (every output as lambda, separately
   divided, leaf included) and leaves c2v[] as LLRs for the LLR-domain form to continue with */
static int fused_cn_pass(dec_t *d, const double *rho, int shared, int handover, uint8_t *lbit, double *ltot)
{
    const spm *H = &d->code->H;
    int escaped = 0;
    for (int i = 0; i < H->rows; ++i)
    {
        const int cw = H->rptr[i + 1] - H->rptr[i];
        const int *cn = H->redge + H->rptr[i], *cols = H->rnode + H->rptr[i];
        int order[4], leaf;
        unsigned flip;
        double v[4], tot = 1.0;
        uint32_t lb = 0, h;
        (void)rho;
        if (!dm_fused_row_order(cw, cols, H->cptr, order, &flip, &leaf))
            return 1; /* cannot happen: fused_applies */
        for (int k = 0; k < cw; ++k)
            v[k] = d->v2c[cn[order[k]]];
        if (handover)
            flip = 0, leaf = 0;
        if (cw == 2)
            h = dm_cnf2(v, flip);
        else if (cw == 3)
            h = dm_cnf3(v, flip, leaf, shared, &lb, &tot);
        else
            h = dm_cnf4(v, flip, leaf, shared, &lb, &tot);
        escaped |= shared && h >= DM_FUSED_P_HI;
        for (int k = 0; k < cw - leaf; ++k)
            d->c2v[cn[order[k]]] = handover ? 0.0 - dm_log(v[k]) : v[k];
        if (leaf)
            lbit[cols[order[cw - 1]]] = (uint8_t)lb, ltot[cols[order[cw - 1]]] = tot;
    }
    return escaped;
}

static int dec_decode_fused(dec_t *d)
{
    const spm *H = &d->code->H;
    const int nc = H->cols;
    double *lam = malloc(8 * (size_t)nc), *rho = malloc(8 * (size_t)nc), *ltot = malloc(8 * (size_t)nc);
    uint8_t *lbit = malloc((size_t)nc);
    int escaped = 0, ret = -1;
    int32_t ho_key = 0;
    for (int i = 0; i < nc; ++i)
    {
        double L = d->llr_in[i];
        escaped |= !(fabs(L) <= DM_RATIO_LLR_LIMIT);
        lam[i] = dm_exp_clamped(0.0 - L);
        rho[i] = dm_exp_clamped(L); /* (the fused form takes both from the exponential: no division in its prologue) */
        for (int p = H->cptr[i]; p < H->cptr[i + 1]; ++p)
            d->v2c[H->cedge[p]] = rho[i];
    }
    /* with early termination: shared reciprocals, and loop pass I = check-node pass I, then the checks on what variable-node
       pass I-1 left behind, then variable-node pass I (dec_decode_ratio).  Without: separately divided outputs, the checks
       come first, and a frame whose totals near the edge of the box is handed to the LLR-domain form (detmath.h "Hand-over") */
    const int shared = d->early_term;
    unsigned I = 0;
    for (;;)
    {
        if (!d->early_term)
        {
            if (escaped)
                break;
            if (I == d->iterations)
            {
                ret = (int)I;
                break;
            }
            if (DM_HANDOVER_DUE(ho_key))
            {
                fused_cn_pass(d, rho, 0, 1, lbit, ltot);
                free(lam), free(rho), free(ltot), free(lbit);
                return dec_decode_llr_from(d, I, 1);
            }
        }
        escaped |= fused_cn_pass(d, rho, shared, 0, lbit, ltot);
        if (d->early_term)
        {
            if (escaped)
                break;
            if (I > 0 && is_codeword(d))
            {
                ret = (int)I - 1;
                break;
            }
            if (I == d->iterations)
            {
                ret = (int)I;
                break;
            }
        }
        for (int i = 0; i < nc; ++i)
        {
            const int deg = H->cptr[i + 1] - H->cptr[i];
            if (deg == 1) /* the decision its check node took in this pass (decoder.cpp:58 for a node of degree 1) */
            {
                d->co[i] = lbit[i];
                d->llr_out[i] = 0.0 - dm_log(ltot[i]);
            }
            else if (deg == 2)
            {
                const int e0 = H->cedge[H->cptr[i]], e1 = H->cedge[H->cptr[i] + 1];
                const double c0 = d->c2v[e0], c1 = d->c2v[e1];
                const double o0 = rho[i] * c1, o1 = rho[i] * c0, tot = o0 * c0;
                const int32_t hk = dm_handover_key(tot);
                ho_key = hk > ho_key ? hk : ho_key;
                d->co[i] = (uint8_t)(tot <= 1.0);
                d->llr_out[i] = dm_log(tot);
                escaped |= dm_ratio_out_of_range(o0) | dm_ratio_out_of_range(o1);
                d->v2c[e0] = o0, d->v2c[e1] = o1;
            }
            else
            {
                double prod = lam[i];
                for (int k = 0; k < deg; ++k)
                {
                    prod *= d->c2v[H->cedge[H->cptr[i] + k]];
                    if (deg > 3 && k % 3 == 2)
                        escaped |= dm_ratio_out_of_range(prod);
                }
                const int32_t hk = dm_handover_key(prod);
                ho_key = hk > ho_key ? hk : ho_key;
                d->co[i] = (uint8_t)(prod >= 1.0);
                d->llr_out[i] = 0.0 - dm_log(prod);
                const double tot = 1.0 / prod;
                for (int k = 0; k < deg; ++k)
                {
                    const int e = H->cedge[H->cptr[i] + k];
                    const double o = tot * d->c2v[e];
                    escaped |= dm_ratio_out_of_range(o);
                    d->v2c[e] = o;
                }
            }
        }
        ++I;
    }
    free(lam), free(rho), free(ltot), free(lbit);
    return ret;
}

static int dec_decode_llr(dec_t *d);

/* test introspection: frames the ratio form finished / handed back since the last reset (not thread-safe) */
static uint64_t g_ratio_done, g_ratio_escaped, g_ratio_second;
void orc_ratio_stats(uint64_t *done, uint64_t *escaped, int reset)
{
    if (done)
        *done = g_ratio_done;
    if (escaped)
        *escaped = g_ratio_escaped;
    if (reset)
        g_ratio_done = g_ratio_escaped = g_ratio_second = 0;
}

/* test introspection: frames whose first (shared-reciprocal) attempt was given up since the last reset of orc_ratio_stats */
uint64_t orc_ratio_second(void) { return g_ratio_second; }

/* decoder.cpp:11-78 */
static int dec_decode(dec_t *d)
{
    if (d->cn == jacobian_det && d->iterations > 0 && (d->early_term || handover_applies(d->code)))
    {
        /* three stages, as the kernels' three launches: shared-reciprocal check nodes; if a value (or a denominator product)
           leaves its range, again from scratch with separately divided outputs; if the box is left there too, the LLR domain */
        /* (first stage: the fused form where the code's structure admits it, fused_rule.h) */
        int it = fused_applies(d->code) ? dec_decode_fused(d) : dec_decode_ratio(d, 1);
        if (it < 0 && d->early_term)
        {
            ++g_ratio_second;
            it = dec_decode_ratio(d, 0);
        }
        if (it >= 0)
        {
            ++g_ratio_done;
            return it;
        }
        ++g_ratio_escaped;
    }
    return dec_decode_llr(d);
}

static int dec_decode_llr(dec_t *d)
{
    const spm *H = &d->code->H;
    for (int e = 0; e < H->nnz; ++e)
        d->v2c[e] = d->llr_in[H->ecol[e]];
    return dec_decode_llr_from(d, 0, 0);
}

static int dec_decode_llr_from(dec_t *d, unsigned first, int skip_first_cn)
{
    const spm *H = &d->code->H;
    unsigned I = first;
    while (I < d->iterations)
    {
        /* CN pass, decoder.cpp:25-45 */
        for (int i = 0; i < H->rows && !(skip_first_cn && I == first); ++i)
        {
            int cw = H->rptr[i + 1] - H->rptr[i];
            const int *cn = H->redge + H->rptr[i];
            double *F = d->F, *B = d->B;
            if (d->cn == jacobian_det && cn_update_det_shared(d, cn, cw))
                continue;
            F[0] = d->v2c[cn[0]];
            B[cw - 1] = d->v2c[cn[cw - 1]];
            for (int j = 1; j < cw; ++j)
            {
                F[j] = d->cn(F[j - 1], d->v2c[cn[j]]);
                B[cw - 1 - j] = d->cn(B[cw - j], d->v2c[cn[cw - j - 1]]);
            }
            d->c2v[cn[0]] = B[1];
            d->c2v[cn[cw - 1]] = F[cw - 2];
            for (int j = 1; j < cw - 1; ++j)
                d->c2v[cn[j]] = d->cn(F[j - 1], B[j + 1]);
        }
        /* VN pass + APP + hard decision, decoder.cpp:48-64 */
        for (int i = 0; i < H->cols; ++i)
        {
            double out = d->llr_in[i];
            for (int p = H->cptr[i]; p < H->cptr[i + 1]; ++p)
                out += d->c2v[H->cedge[p]];
            d->llr_out[i] = out;
            d->co[i] = (uint8_t)(out <= 0);
            for (int p = H->cptr[i]; p < H->cptr[i + 1]; ++p)
                d->v2c[H->cedge[p]] = out - d->c2v[H->cedge[p]];
        }
        if (d->early_term && is_codeword(d)) /* decoder.cpp:66-72 */
            break;
        ++I;
    }
    return (int)I;
}

int orc_decode(const orc_code *c, int min_sum, int early_term, unsigned iterations, int math_mode,
               const double *llr_in, double *llr_out, uint8_t *hard)
{
    dec_t d;
    dec_init(&d, c, min_sum, early_term, iterations, math_mode);
    memcpy(d.llr_in, llr_in, 8 * (size_t)c->H.cols);
    int it = dec_decode(&d);
    if (llr_out)
        memcpy(llr_out, d.llr_out, 8 * (size_t)c->H.cols);
    if (hard)
        memcpy(hard, d.co, c->H.cols);
    dec_free(&d);
    return it;
}

/* ------------------------------------------------------------------------------------------------ */
/* Mirrors of the opt-in NON-PARITY modes (libldpc_amd/csrc/kernels_fast.hip,     */
/* kernels_layered.hip).  Nothing here is the reference's algorithm — the reference has neither     */
/* binary32 messages nor a layered schedule — so there is no fixture to pin these against; they     */
/* restate, independently and in plain C with libm's exp2f / log2f and an IEEE 1.0f / x where the   */
/* kernels use v_exp_f32 / v_log_f32 / v_rcp_f32, the schedule and the arithmetic the kernels claim */
/* to implement.  The hardware instructions are accurate to about one ulp, not correctly rounded:   */
/* tests compare with tolerances (same iteration / sweep counts and decisions on >= 99.9 % of the   */
/* frames, LLRs to 1e-3 relative on converged frames), never bit for bit.                           */
/* ------------------------------------------------------------------------------------------------ */
#define FAST_CLIP 40.0f
#define FAST_LOG2E 1.4426950408889634f
static inline float fclip(float x) { return fminf(fmaxf(x, -FAST_CLIP), FAST_CLIP); }
static inline float f_lambda(float a, float b) { return (a + b) * (1.0f / fmaf(a, b, 1.0f)); }
static inline float f_rho(float a, float b) { return fmaf(a, b, 1.0f) * (1.0f / (a + b)); }

/* check node on D likelihood ratios v[] (rho = 2^L2): lambda(c2v_j) out, forward/backward order of decoder.cpp:31-44 */
static void fast_cn(int D, const float *v, float *o)
{
    float F[64], B[64];
    if (D == 2)
    {
        o[0] = 1.0f / v[1], o[1] = 1.0f / v[0];
        return;
    }
    F[0] = v[0], B[D - 1] = v[D - 1];
    for (int j = 1; j <= D - 3; ++j)
        F[j] = f_rho(F[j - 1], v[j]);
    for (int j = D - 2; j >= 2; --j)
        B[j] = f_rho(B[j + 1], v[j]);
    o[0] = f_lambda(D > 3 ? B[2] : v[2], v[1]);
    o[D - 1] = f_lambda(D > 3 ? F[D - 3] : v[0], v[D - 2]);
    for (int j = 1; j < D - 1; ++j)
        o[j] = f_lambda(F[j - 1], B[j + 1]);
}

/* mode 1: flooding sum-product, binary32 messages (v2c as rho = 2^L2 with the sender's decision in the sign, c2v as
   lambda = 2^-L2), L2 = LLR in log2 units clipped to +-40; the variable node sums log2 values */
static int dec_decode_fast32(const orc_code *c, int early_term, unsigned iterations, const double *llr_in, double *llr_out, uint8_t *hard)
{
    const spm *H = &c->H;
    const int nc = H->cols, nnz = H->nnz;
    float *msg = malloc(4 * (size_t)(nnz > 0 ? nnz : 1)), *l2 = malloc(4 * (size_t)nc);
    uint8_t *dec = calloc((size_t)nnz + 1, 1), *co = calloc((size_t)nc, 1);
    for (int i = 0; i < nc; ++i)
    {
        l2[i] = fclip((float)llr_in[i] * FAST_LOG2E);
        for (int p = H->cptr[i]; p < H->cptr[i + 1]; ++p)
            msg[H->cedge[p]] = exp2f(l2[i]), dec[H->cedge[p]] = 0;
        if (llr_out)
            llr_out[i] = 0.0;
    }
    unsigned I = 0;
    for (;;)
    {
        int bad = 0;
        for (int i = 0; i < H->rows; ++i)
        {
            const int cw = H->rptr[i + 1] - H->rptr[i];
            const int *cn = H->redge + H->rptr[i];
            float v[64], o[64];
            int par = 0;
            for (int j = 0; j < cw; ++j)
                v[j] = msg[cn[j]], par ^= dec[cn[j]];
            bad |= par;
            fast_cn(cw, v, o);
            for (int j = 0; j < cw; ++j)
                msg[cn[j]] = o[j];
        }
        if (I > 0 && early_term && !bad)
        {
            --I;
            break;
        }
        if (I == iterations)
            break;
        for (int i = 0; i < nc; ++i)
        {
            const int deg = H->cptr[i + 1] - H->cptr[i];
            float cl[64], tot = l2[i];
            for (int k = 0; k < deg; ++k)
            {
                cl[k] = log2f(msg[H->cedge[H->cptr[i] + k]]);
                tot -= cl[k];
            }
            co[i] = (uint8_t)(tot <= 0.0f);
            if (llr_out)
                llr_out[i] = (double)tot * 0.6931471805599453;
            for (int k = 0; k < deg; ++k)
            {
                const int e = H->cedge[H->cptr[i] + k];
                msg[e] = exp2f(fclip(tot + cl[k]));
                dec[e] = co[i];
            }
        }
        ++I;
    }
    if (hard)
        for (int i = 0; i < nc; ++i)
            hard[i] = iterations > 0 ? co[i] : 0;
    free(msg), free(l2), free(dec), free(co);
    return (int)I;
}

/* binary32 -> binary16 -> binary32, round to nearest even (what a store of a _Float16 message and its reload do); |x| <= 40 */
static float through_half(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7FFFFFFFu;
    float r;
    if (a < 0x38800000u) /* below 2^-14: a binary16 subnormal, multiples of 2^-24 */
    {
        float ax;
        memcpy(&ax, &a, 4);
        const float q = ax * 16777216.0f; /* exact scaling; rintf rounds to nearest even */
        r = rintf(q) * (1.0f / 16777216.0f);
    }
    else
    {
        const uint32_t keep = a & 0xFFFFE000u, rest = a & 0x1FFFu; /* 13 bits dropped */
        uint32_t o = keep;
        if (rest > 0x1000u || (rest == 0x1000u && (keep & 0x2000u)))
            o += 0x2000u;
        memcpy(&r, &o, 4);
    }
    uint32_t ru;
    memcpy(&ru, &r, 4);
    ru |= sign;
    memcpy(&r, &ru, 4);
    return r;
}

/* Steps of the layered schedule (libldpc_amd/csrc/plan.cpp, build_layer_plan, restated): rows in file order, each put
   into the first step of its degree that has a free lane (64 per step) and none of its variable nodes yet; steps in the
   order they were opened.  step_of[row] = step index; returns the number of steps. */
int orc_layer_steps(const orc_code *c, int *step_of)
{
    const spm *H = &c->H;
    int n_steps = 0, cap = 16;
    int *deg = malloc(sizeof(int) * cap), *cnt = malloc(sizeof(int) * cap);
    uint8_t **used = malloc(sizeof(uint8_t *) * cap);
    for (int i = 0; i < H->rows; ++i)
    {
        const int d = H->rptr[i + 1] - H->rptr[i];
        int dst = -1;
        for (int s = 0; s < n_steps && dst < 0; ++s)
        {
            if (deg[s] != d || cnt[s] >= 64)
                continue;
            int clash = 0;
            for (int p = H->rptr[i]; p < H->rptr[i + 1] && !clash; ++p)
                clash = used[s][H->rnode[p]] != 0;
            if (!clash)
                dst = s;
        }
        if (dst < 0)
        {
            if (n_steps == cap)
            {
                cap *= 2;
                deg = realloc(deg, sizeof(int) * cap), cnt = realloc(cnt, sizeof(int) * cap);
                used = realloc(used, sizeof(uint8_t *) * cap);
            }
            dst = n_steps++;
            deg[dst] = d, cnt[dst] = 0, used[dst] = calloc((size_t)H->cols, 1);
        }
        ++cnt[dst];
        step_of[i] = dst;
        for (int p = H->rptr[i]; p < H->rptr[i + 1]; ++p)
            used[dst][H->rnode[p]] = 1;
    }
    for (int s = 0; s < n_steps; ++s)
        free(used[s]);
    free(used), free(deg), free(cnt);
    return n_steps;
}

/* modes 2 / 3: layered (row-serial) sum-product, binary32 totals, binary32 (half = 0) or binary16 (half = 1) check-to-
   variable messages in log2 units; returns the sweeps completed before the sweep whose syndrome check passed */
static int dec_decode_layered(const orc_code *c, int half, int early_term, unsigned iterations, const double *llr_in, double *llr_out,
                              uint8_t *hard)
{
    const spm *H = &c->H;
    const int nc = H->cols, mc = H->rows;
    int *step_of = malloc(sizeof(int) * (size_t)(mc > 0 ? mc : 1));
    const int n_steps = orc_layer_steps(c, step_of);
    /* rows in processing order: step by step (the order inside a step does not matter: its rows share no variable node) */
    int *order = malloc(sizeof(int) * (size_t)(mc > 0 ? mc : 1)), k = 0;
    for (int s = 0; s < n_steps; ++s)
        for (int i = 0; i < mc; ++i)
            if (step_of[i] == s)
                order[k++] = i;
    float *tot = malloc(4 * (size_t)nc), *c2v = calloc((size_t)(H->nnz > 0 ? H->nnz : 1), 4);
    for (int i = 0; i < nc; ++i)
        tot[i] = fclip((float)llr_in[i] * FAST_LOG2E);
    unsigned I = 0;
    while (I < iterations)
    {
        for (int q = 0; q < mc; ++q)
        {
            const int i = order[q], cw = H->rptr[i + 1] - H->rptr[i];
            const int *cn = H->redge + H->rptr[i], *col = H->rnode + H->rptr[i];
            float t[64], v[64], o[64];
            for (int j = 0; j < cw; ++j)
            {
                t[j] = tot[col[j]] - c2v[cn[j]];
                v[j] = exp2f(fclip(t[j]));
            }
            fast_cn(cw, v, o);
            for (int j = 0; j < cw; ++j)
            {
                float m = fclip(0.0f - log2f(o[j]));
                if (half)
                    m = through_half(m);
                c2v[cn[j]] = m;
                tot[col[j]] = t[j] + m;
            }
        }
        if (early_term)
        {
            int bad = 0;
            for (int i = 0; i < mc && !bad; ++i)
            {
                int par = 0;
                for (int p = H->rptr[i]; p < H->rptr[i + 1]; ++p)
                    par ^= tot[H->rnode[p]] <= 0.0f;
                bad |= par;
            }
            if (!bad)
                break;
        }
        ++I;
    }
    for (int i = 0; i < nc; ++i)
    {
        if (hard)
            hard[i] = iterations > 0 ? (uint8_t)(tot[i] <= 0.0f) : 0;
        if (llr_out)
            llr_out[i] = iterations > 0 ? (double)tot[i] * 0.6931471805599453 : 0.0;
    }
    free(step_of), free(order), free(tot), free(c2v);
    return (int)I;
}

/* n frames (llr_in[n][nc], column order) through the mirror of fast mode `mode` (1: flooding binary32, 2: layered binary32,
   3: layered binary16 messages), OpenMP over frames */
void orc_decode_fast_batch(const orc_code *c, int mode, int early_term, unsigned iterations, uint64_t n, const double *llr_in,
                           uint32_t *iters, double *llr_out, uint8_t *hard)
{
    const size_t nc = (size_t)c->H.cols;
#pragma omp parallel for schedule(dynamic, 16)
    for (uint64_t f = 0; f < n; ++f)
    {
        double *lo = llr_out ? llr_out + f * nc : NULL;
        uint8_t *hd = hard ? hard + f * nc : NULL;
        const int it = mode == 1 ? dec_decode_fast32(c, early_term, iterations, llr_in + f * nc, lo, hd)
                                 : dec_decode_layered(c, mode == 3, early_term, iterations, llr_in + f * nc, lo, hd);
        if (iters)
            iters[f] = (uint32_t)it;
    }
}

/* ------------------------------------------------------------------------------------ */
/* BEC decoder (decoder.h:130-156, decoder.cpp:91-192)                                   */
/* ------------------------------------------------------------------------------------ */
typedef struct
{
    const orc_code *code;
    int early_term, deg1_compat;
    unsigned iterations;
    uint8_t *v2c, *c2v, *F, *B, *llr_in, *llr_out, *co;
} becdec_t;

static inline uint8_t bec_vn(uint8_t l, uint8_t r, uint8_t xi) /* decoder.h:145-148 */
{
    return (xi == l || xi == r) ? xi : ORC_ERASURE;
}
static inline uint8_t bec_cn(uint8_t l, uint8_t r) /* decoder.h:152-155 */
{
    return (l == ORC_ERASURE || r == ORC_ERASURE) ? ORC_ERASURE : (uint8_t)((l != 0) ^ (r != 0));
}

static void becdec_init(becdec_t *d, const orc_code *c, int early_term, unsigned iters, int compat)
{
    d->code = c, d->early_term = early_term, d->iterations = iters, d->deg1_compat = compat;
    int nnz = c->H.nnz, nc = c->H.cols, md = c->max_degree > 2 ? c->max_degree : 2;
    d->v2c = calloc(nnz, 1), d->c2v = calloc(nnz, 1);
    d->F = calloc(md, 1), d->B = calloc(md, 1);
    d->llr_in = calloc(nc, 1), d->llr_out = calloc(nc, 1), d->co = calloc(nc, 1);
}
static void becdec_free(becdec_t *d)
{
    free(d->v2c), free(d->c2v), free(d->F), free(d->B), free(d->llr_in), free(d->llr_out), free(d->co);
}

static int becdec_decode(becdec_t *d, const uint8_t *x /* true codeword, gf2 values */)
{
    const spm *H = &d->code->H;
    for (int e = 0; e < H->nnz; ++e)
        d->v2c[e] = d->llr_in[H->ecol[e]];
    unsigned I = 0;
    while (I < d->iterations)
    {
        for (int i = 0; i < H->rows; ++i) /* decoder.cpp:105-123 */
        {
            int cw = H->rptr[i + 1] - H->rptr[i];
            const int *cn = H->redge + H->rptr[i];
            uint8_t *F = d->F, *B = d->B;
            F[0] = d->v2c[cn[0]];
            B[cw - 1] = d->v2c[cn[cw - 1]];
            for (int j = 1; j < cw; ++j)
            {
                F[j] = bec_cn(F[j - 1], d->v2c[cn[j]]);
                B[cw - 1 - j] = bec_cn(B[cw - j], d->v2c[cn[cw - j - 1]]);
            }
            d->c2v[cn[0]] = B[1];
            d->c2v[cn[cw - 1]] = F[cw - 2];
            for (int j = 1; j < cw - 1; ++j)
                d->c2v[cn[j]] = bec_cn(F[j - 1], B[j + 1]);
        }
        for (int i = 0; i < H->cols; ++i) /* decoder.cpp:126-167 */
        {
            int vw = H->cptr[i + 1] - H->cptr[i];
            const int *vn = H->cedge + H->cptr[i];
            if (d->llr_in[i] != ORC_ERASURE)
            {
                for (int p = 0; p < vw; ++p)
                    d->v2c[vn[p]] = x[i];
                d->llr_out[i] = x[i];
                d->co[i] = x[i];
            }
            else
            {
                uint8_t *F = d->F, *B = d->B;
                if (vw == 0)
                {
                    /* isolated erased VN: the reference indexes an empty neighbour list (undefined);
                       defined here and in the kernel as "stays erased" */
                    d->llr_out[i] = ORC_ERASURE;
                }
                else if (vw == 1)
                {
                    /* The reference reads mExMsgF[-1] here (decoder.cpp:155-156).
                       deg1_compat reproduces what that read returns with glibc malloc (0);
                       otherwise the defined extrinsic of a degree-1 erased VN: an erasure. */
                    F[0] = d->c2v[vn[0]];
                    d->v2c[vn[0]] = d->deg1_compat ? 0 : ORC_ERASURE;
                    d->llr_out[i] = F[0];
                }
                else
                {
                    F[0] = d->c2v[vn[0]];
                    B[vw - 1] = d->c2v[vn[vw - 1]];
                    for (int j = 1; j < vw; ++j)
                    {
                        F[j] = bec_vn(F[j - 1], d->c2v[vn[j]], x[i]);
                        B[vw - 1 - j] = bec_vn(B[vw - j], d->c2v[vn[vw - j - 1]], x[i]);
                    }
                    d->v2c[vn[0]] = B[1];
                    d->v2c[vn[vw - 1]] = F[vw - 2];
                    for (int j = 1; j < vw - 1; ++j)
                        d->v2c[vn[j]] = bec_vn(F[j - 1], B[j + 1], x[i]);
                    d->llr_out[i] = F[vw - 1];
                }
                /* -gf2 is always 1 (gf2.cpp:5-8) */
                d->co[i] = (d->llr_out[i] == ORC_ERASURE) ? 1 : x[i];
            }
        }
        if (d->early_term) /* decoder.cpp:169-186 */
        {
            int found = 0;
            for (int i = 0; i < H->cols; ++i)
                if (d->llr_out[i] == ORC_ERASURE)
                {
                    found = 1;
                    break;
                }
            if (!found)
                break;
        }
        ++I;
    }
    return (int)I;
}

int orc_decode_bec(const orc_code *c, int early_term, unsigned iterations, int deg1_compat,
                   const uint8_t *llr_in, const uint8_t *codeword, uint8_t *llr_out, uint8_t *hard)
{
    becdec_t d;
    becdec_init(&d, c, early_term, iterations, deg1_compat);
    memcpy(d.llr_in, llr_in, c->H.cols);
    int it = becdec_decode(&d, codeword);
    if (llr_out)
        memcpy(llr_out, d.llr_out, c->H.cols);
    if (hard)
        memcpy(hard, d.co, c->H.cols);
    becdec_free(&d);
    return it;
}

/* ------------------------------------------------------------------------------------ */
/* libstdc++ <random> restated                                                           */
/* ------------------------------------------------------------------------------------ */
#define MT_N 312
#define MT_M 156
typedef struct
{
    uint64_t x[MT_N];
    int p;
    uint64_t draws;
} mt64;

static void mt64_seed(mt64 *g, uint64_t seed)
{
    g->x[0] = seed;
    for (int i = 1; i < MT_N; ++i)
        g->x[i] = 6364136223846793005ull * (g->x[i - 1] ^ (g->x[i - 1] >> 62)) + (uint64_t)i;
    g->p = MT_N;
    g->draws = 0;
}

static uint64_t mt64_next(mt64 *g)
{
    if (g->p >= MT_N)
    {
        const uint64_t UM = 0xFFFFFFFF80000000ull, LM = 0x7FFFFFFFull, A = 0xB5026F5AA96619E9ull;
        for (int k = 0; k < MT_N; ++k)
        {
            uint64_t y = (g->x[k] & UM) | (g->x[(k + 1) % MT_N] & LM);
            g->x[k] = g->x[(k + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1) ? A : 0);
        }
        g->p = 0;
    }
    uint64_t z = g->x[g->p++];
    z ^= (z >> 29) & 0x5555555555555555ull;
    z ^= (z << 17) & 0x71D67FFFEDA60000ull;
    z ^= (z << 37) & 0xFFF7EEE000000000ull;
    z ^= (z >> 43);
    g->draws++;
    return z;
}

void orc_mt64_stream(uint64_t seed, uint64_t n, uint64_t *out)
{
    mt64 g;
    mt64_seed(&g, seed);
    for (uint64_t i = 0; i < n; ++i)
        out[i] = mt64_next(&g);
}

/* generate_canonical<double,53>: one draw, u64 -> double (round to nearest) / 2^64 */
static double canonical(mt64 *g)
{
    double r = (double)mt64_next(g) / 18446744073709551616.0;
    if (r >= 1.0)
        r = nextafter(1.0, 0.0);
    return r;
}

typedef struct
{
    double stddev;
    double saved;
    int saved_ok;
} normal_t;

/* normal_distribution::operator() (random.tcc): polar method, returns y*mult first */
static double normal_draw(normal_t *n, mt64 *g, int math)
{
    double ret;
    if (n->saved_ok)
    {
        n->saved_ok = 0;
        ret = n->saved;
    }
    else
    {
        double x, y, r2;
        do
        {
            x = 2.0 * canonical(g) - 1.0;
            y = 2.0 * canonical(g) - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        double mult = sqrt(-2 * orc_log(math, r2) / r2);
        n->saved = x * mult;
        n->saved_ok = 1;
        ret = y * mult;
    }
    return ret * n->stddev + 0.0;
}

static int bernoulli_draw(mt64 *g, double p) { return canonical(g) < p; }

/* ------------------------------------------------------------------------------------ */
/* channels (channel.h, channel.cpp)                                                     */
/* ------------------------------------------------------------------------------------ */
struct orc_chan
{
    const orc_code *code;
    int type, math;
    mt64 seed_state; /* mRNG: never advances (channel.cpp:37-42 binds a COPY)            */
    mt64 noise;      /* the bound copy that actually produces the noise                  */
    mt64 info;       /* mt19937_64(seed << 1), persists across channel points            */
    normal_t nrm;
    double param, sigma2;
    uint8_t *info_word, *cw; /* kc, nc                                                   */
    double *xd, *yd;         /* AWGN channel i/o (nct)                                   */
    uint8_t *xb, *yb;        /* BSC/BEC channel i/o (nct)                                */
    dec_t dec;
    becdec_t bdec;
    double *bec_llr_in_d, *bec_llr_out_d;
};

orc_chan *orc_chan_new(const orc_code *c, int type, uint64_t seed, int math, int min_sum,
                       int early_term, unsigned iterations, int bec_compat)
{
    orc_chan *ch = calloc(1, sizeof *ch);
    ch->code = c, ch->type = type, ch->math = math;
    mt64_seed(&ch->seed_state, seed);
    ch->noise = ch->seed_state;
    mt64_seed(&ch->info, seed << 1); /* channel.cpp:11 */
    int nct = c->nct, nc = c->H.cols;
    ch->info_word = calloc(c->kc > 0 ? c->kc : 1, 1);
    ch->cw = calloc(nc, 1);
    ch->xd = malloc(8 * (size_t)nct), ch->yd = calloc(nct, 8);
    for (int i = 0; i < nct; ++i)
        ch->xd[i] = 1.0; /* all-zero codeword, channel.cpp:28 */
    ch->xb = calloc(nct, 1), ch->yb = calloc(nct, 1);
    if (type == ORC_BEC)
        becdec_init(&ch->bdec, c, early_term, iterations, bec_compat);
    else
        dec_init(&ch->dec, c, min_sum, early_term, iterations, math);
    /* constructor defaults (ldpcsim.cpp:39,52,65): snr 1.0 / eps 0.0; start() always
       calls set_channel_param before the first frame */
    orc_chan_set_param(ch, type == ORC_AWGN ? 1.0 : 0.0);
    return ch;
}

void orc_chan_free(orc_chan *ch)
{
    if (!ch)
        return;
    free(ch->info_word), free(ch->cw), free(ch->xd), free(ch->yd), free(ch->xb), free(ch->yb);
    if (ch->type == ORC_BEC)
        becdec_free(&ch->bdec);
    else
        dec_free(&ch->dec);
    free(ch->bec_llr_in_d), free(ch->bec_llr_out_d);
    free(ch);
}

void orc_chan_set_param(orc_chan *ch, double x)
{
    ch->param = x;
    ch->noise = ch->seed_state; /* std::bind copies mRNG: the stream restarts */
    ch->noise.draws = 0;
    if (ch->type == ORC_AWGN)
    {
        ch->sigma2 = pow(10, -x / 10);
        ch->nrm.stddev = sqrt(ch->sigma2);
        ch->nrm.saved_ok = 0;
    }
}

void orc_chan_encode_and_map(orc_chan *ch)
{
    const orc_code *c = ch->code;
    for (int i = 0; i < c->kc; ++i)
        ch->info_word[i] = (uint8_t)bernoulli_draw(&ch->info, 0.5);
    orc_encode_accumulate(c, ch->info_word, ch->cw);
    for (int i = 0; i < c->nct; ++i)
    {
        uint8_t b = ch->cw[c->bit_pos[i]];
        if (ch->type == ORC_AWGN)
            ch->xd[i] = 1 - (2 * b);
        else
            ch->xb[i] = b;
    }
}

void orc_chan_simulate(orc_chan *ch)
{
    int nct = ch->code->nct;
    if (ch->type == ORC_AWGN)
        for (int i = 0; i < nct; ++i)
            ch->yd[i] = normal_draw(&ch->nrm, &ch->noise, ch->math) + ch->xd[i];
    else if (ch->type == ORC_BSC)
        for (int i = 0; i < nct; ++i)
            ch->yb[i] = ch->xb[i] ^ (uint8_t)bernoulli_draw(&ch->noise, ch->param);
    else
        for (int i = 0; i < nct; ++i)
            ch->yb[i] = bernoulli_draw(&ch->noise, ch->param) ? ORC_ERASURE : ch->xb[i];
}

void orc_chan_calc_llrs(orc_chan *ch)
{
    const orc_code *c = ch->code;
    if (ch->type == ORC_AWGN)
    {
        double *L = ch->dec.llr_in;
        for (int i = 0; i < c->npunct; ++i)
            L[c->punct[i]] = 0.0;
        for (int i = 0; i < c->nshort; ++i)
            L[c->shrt[i]] = 99999.9;
        for (int i = 0; i < c->nct; ++i)
            L[c->bit_pos[i]] = 2 * ch->yd[i] / ch->sigma2;
    }
    else if (ch->type == ORC_BSC)
    {
        double *L = ch->dec.llr_in;
        const double delta = log((1 - ch->param) / ch->param);
        for (int i = 0; i < c->npunct; ++i)
            L[c->punct[i]] = 0.0;
        for (int i = 0; i < c->nshort; ++i)
            L[c->shrt[i]] = delta;
        for (int i = 0; i < c->nct; ++i)
            L[c->bit_pos[i]] = delta * (1 - 2 * (int)ch->yb[i]);
    }
    else
    {
        uint8_t *L = ch->bdec.llr_in;
        for (int i = 0; i < c->npunct; ++i)
            L[c->punct[i]] = ORC_ERASURE;
        for (int i = 0; i < c->nshort; ++i) /* channel.cpp:222: indexes mX by column index */
            L[c->shrt[i]] = c->shrt[i] < c->nct ? ch->xb[c->shrt[i]] : 0;
        for (int i = 0; i < c->nct; ++i)
            L[c->bit_pos[i]] = ch->yb[i];
    }
}

int orc_chan_decode(orc_chan *ch)
{
    return ch->type == ORC_BEC ? becdec_decode(&ch->bdec, ch->cw) : dec_decode(&ch->dec);
}

const double *orc_chan_llr_in(const orc_chan *ch) { return ch->dec.llr_in; }
const double *orc_chan_llr_out(const orc_chan *ch) { return ch->dec.llr_out; }
const uint8_t *orc_chan_llr_in_bec(const orc_chan *ch) { return ch->bdec.llr_in; }
const uint8_t *orc_chan_llr_out_bec(const orc_chan *ch) { return ch->bdec.llr_out; }
const uint8_t *orc_chan_estimate(const orc_chan *ch)
{
    return ch->type == ORC_BEC ? ch->bdec.co : ch->dec.co;
}
const uint8_t *orc_chan_codeword(const orc_chan *ch) { return ch->cw; }
uint64_t orc_chan_raw_draws(const orc_chan *ch) { return ch->noise.draws; }

static uint32_t count_bit_errors(const orc_chan *ch) /* ldpcsim.cpp:184-188 */
{
    const orc_code *c = ch->code;
    const uint8_t *est = orc_chan_estimate(ch);
    uint32_t n = 0;
    for (int i = 0; i < c->n_bitpos; ++i) /* the reference iterates the whole bit_pos vector */
        n += est[c->bit_pos[i]] != ch->cw[c->bit_pos[i]];
    return n;
}

void orc_chan_run_frames(orc_chan *ch, uint64_t skip, uint64_t count, uint32_t *iters,
                         uint32_t *bit_errors, uint8_t *hard, double *llr_in, double *llr_out,
                         uint8_t *codeword)
{
    const orc_code *c = ch->code;
    size_t nc = (size_t)c->H.cols;
    for (uint64_t f = 0; f < skip + count; ++f)
    {
        if (c->hasG)
            orc_chan_encode_and_map(ch);
        orc_chan_simulate(ch);
        if (f < skip)
            continue;
        orc_chan_calc_llrs(ch);
        uint64_t k = f - skip;
        if (llr_in)
            for (size_t i = 0; i < nc; ++i)
                llr_in[k * nc + i] = ch->type == ORC_BEC ? (double)ch->bdec.llr_in[i] : ch->dec.llr_in[i];
        int it = orc_chan_decode(ch);
        if (iters)
            iters[k] = (uint32_t)it;
        if (bit_errors)
            bit_errors[k] = count_bit_errors(ch);
        if (hard)
            memcpy(hard + k * nc, orc_chan_estimate(ch), nc);
        if (llr_out)
            for (size_t i = 0; i < nc; ++i)
                llr_out[k * nc + i] = ch->type == ORC_BEC ? (double)ch->bdec.llr_out[i] : ch->dec.llr_out[i];
        if (codeword)
            memcpy(codeword + k * nc, ch->cw, nc);
    }
}

/* ------------------------------------------------------------------------------------ */
/* simulation loop (ldpcsim.cpp:97-263)                                                  */
/* ------------------------------------------------------------------------------------ */
static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int orc_simulate(const orc_code *c, int chan_type, uint64_t seed, const double xr[3], int min_sum,
                 int early_term, unsigned iterations, int math, int bec_compat, unsigned threads,
                 uint64_t max_frames, uint64_t min_fec, orc_results *res, uint64_t *totals,
                 const volatile uint8_t *stop_flag)
{
    /* x values: ldpcsim.cpp:104-122 */
    int nx = 0, capx = 0;
    double *xv = NULL;
    double val = xr[0];
    while (val < xr[1])
    {
        if (nx == capx)
            xv = realloc(xv, 8 * (size_t)(capx = capx ? capx * 2 : 64));
        xv[nx++] = val;
        val += xr[2];
    }
    if (chan_type == ORC_BSC || chan_type == ORC_BEC)
        for (int i = 0; i < nx / 2; ++i)
        {
            double t = xv[i];
            xv[i] = xv[nx - 1 - i];
            xv[nx - 1 - i] = t;
        }
    if (threads < 1)
        threads = 1;
    orc_chan **chs = calloc(threads, sizeof *chs);
    for (unsigned t = 0; t < threads; ++t) /* ldpcsim.cpp:29-75: seed + i */
        chs[t] = orc_chan_new(c, chan_type, seed + t, math, min_sum, early_term, iterations, bec_compat);
    static const volatile uint8_t never = 0;
    if (!stop_flag)
        stop_flag = &never;
    const uint64_t nc = (uint64_t)c->H.cols;

    for (int i = 0; i < nx; ++i)
    {
        uint64_t bec = 0, fec = 0, frames = 0, iters = 0;
        double t_start = now_us();
#pragma omp parallel num_threads(threads) shared(bec, fec, frames, iters, t_start)
        {
#ifdef _OPENMP
            unsigned tid = (unsigned)omp_get_thread_num();
#else
            unsigned tid = 0;
#endif
            orc_chan *ch = chs[tid];
            orc_chan_set_param(ch, xv[i]);
            uint64_t fec_seen, frames_seen;
            do
            {
                if (c->hasG)
                    orc_chan_encode_and_map(ch);
                orc_chan_simulate(ch);
                orc_chan_calc_llrs(ch);
                uint64_t it = (uint64_t)orc_chan_decode(ch);
#pragma omp atomic update
                iters += it;
#pragma omp atomic read
                fec_seen = fec;
                if (fec_seen < min_fec)
                {
                    uint64_t fr;
#pragma omp atomic capture
                    fr = ++frames;
                    uint32_t bt = count_bit_errors(ch);
                    if (bt > 0)
                    {
                        double t_now = now_us();
                        uint64_t t_frame = (uint64_t)(t_now - t_start) / fr;
#pragma omp critical
                        {
                            bec += bt;
                            ++fec;
                            if (res)
                            {
                                res->fer[i] = (double)fec / frames;
                                res->ber[i] = (double)bec / (frames * nc);
                                res->avg_iter[i] = (double)iters / frames;
                                res->time[i] = (double)t_frame * 1e-6;
                                res->fec[i] = fec;
                                res->frames[i] = frames;
                            }
                        }
                    }
                }
#pragma omp atomic read
                fec_seen = fec;
#pragma omp atomic read
                frames_seen = frames;
            } while (fec_seen < min_fec && frames_seen < max_frames && !*stop_flag);
        }
        if (totals)
        {
            totals[4 * i + 0] = frames, totals[4 * i + 1] = fec;
            totals[4 * i + 2] = bec, totals[4 * i + 3] = iters;
        }
    }
    for (unsigned t = 0; t < threads; ++t)
        orc_chan_free(chs[t]);
    free(chs);
    free(xv);
    return nx;
}
