/*
 * ps_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED).
 * See ps_oracle.h for the status statement.  Every stage names the upstream
 * lh3/bwa 0.7.x function whose published behaviour it restates; the caller in
 * the reference tree is PARAsuiteMapping.java:63-92 / BWAMapping.java:51-75.
 * Written from the algorithm's description, in this repository's own layout:
 * no upstream or reference source is included.
 */
#define _GNU_SOURCE
#include "ps_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[512];
const char *orc_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); } while (0)

/* ------------------------------------------------------------------ RNG -- */
/* POSIX 48-bit LCG: X' = (0x5DEECE66D * X + 0xB) mod 2^48; srand48 sets the
 * high 32 bits to the seed and the low 16 to 0x330E.  Upstream uses
 * srand48(11)+lrand48() for N bases (bntseq.c) and srand48(bns->seed=11)+
 * drand48() for the tie-break in samse (bwase.c: bwa_aln2seq_core). */
void orc_srand48(orc_rng_t *r, long seed) { r->x = (((uint64_t)(uint32_t)seed) << 16) | 0x330E; }
static inline uint64_t rng_step(orc_rng_t *r) {
    r->x = (r->x * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL;
    return r->x;
}
double orc_drand48(orc_rng_t *r) { return (double)rng_step(r) * (1.0 / 281474976710656.0); }
long   orc_lrand48(orc_rng_t *r) { return (long)(rng_step(r) >> 17); }

/* -------------------------------------------------------------- options -- */
void orc_default_opt(orc_opt_t *o) /* upstream gap_init_opt (bwtaln.c) */
{
    memset(o, 0, sizeof *o);
    o->max_diff = -1; o->fnr = 0.04;
    o->max_gapo = 1; o->max_gape = 6; o->mode_gape = 1;
    o->indel_end_skip = 5; o->max_del_occ = 10; o->max_entries = 2000000;
    o->seed_len = 32; o->max_seed_diff = 2; o->max_top2 = 30;
    o->s_mm = 3; o->s_gapo = 11; o->s_gape = 4;
    o->n_occ = 3;
    o->profile = 0; o->unit = 1; o->x_avg_mm = -1;
}

/* upstream bwa_cal_maxdiff (bwtaln.c): smallest k with Poisson tail < thres */
int orc_cal_maxdiff(int l, double err, double thres)
{
    double elambda = exp(-l * err), sum, y = 1.0;
    int k, x = 1;
    for (k = 1, sum = elambda; k < 1000; ++k) {
        y *= l * err;
        x = (int)((unsigned)x * (unsigned)k);
        sum += elambda * y / x;
        if (1.0 - sum < thres) return k;
    }
    return 2;
}

int orc_mapq_logn(int n) { return (int)(4.343 * log((double)n) + 0.5); } /* upstream g_log_n */

/*
 * OUR cost model for `bwa parasuite -p EP -g IP -X x` (the fork's rule is not
 * available).  P[a][b] = P(read b | ref a).  With
 * Lbar = mean over the 12 off-diagonal entries of ln P, one "average mismatch"
 * costs U units and
 *     cost(a->b) = clamp(round(U * ln P[a][b] / Lbar), 1, 4U)
 * so frequent PAR-CLIP T->C conversions are cheap and rare substitutions dear.
 * Gap open: clamp(round(U * ln rate / Lbar), U, 8U) per type, stock ratio
 * 11/3*U when the rate is absent; gap extension round(4/3*U).  Budget =
 * x * U (x<0: the stock per-length table times U).
 */
void orc_profile_costs(orc_opt_t *o, const double P[16], double ins_rate, double del_rate, int x_avg_mm)
{
    const int U = 8;
    double L[16], lbar = 0.0;
    int a, b;
    for (a = 0; a < 4; ++a) for (b = 0; b < 4; ++b) {
        double p = P[a * 4 + b];
        if (!(p > 1e-9)) p = 1e-9;       /* also catches NaN */
        if (p > 1.0) p = 1.0;
        L[a * 4 + b] = log(p);
        if (a != b) lbar += L[a * 4 + b];
    }
    lbar /= 12.0;
    if (lbar > -1e-6) lbar = -1e-6;
    o->profile = 1; o->unit = U; o->x_avg_mm = x_avg_mm;
    for (a = 0; a < 4; ++a) for (b = 0; b < 4; ++b) {
        int c = 0;
        if (a != b) {
            c = (int)floor(U * L[a * 4 + b] / lbar + 0.5);
            if (c < 1) c = 1;
            if (c > 4 * U) c = 4 * U;
        }
        o->sub_cost[a * 4 + b] = c;
    }
    o->n_cost = U;
    {
        double r[2] = { ins_rate, del_rate };
        int32_t *dst[2] = { &o->gapo_ins_cost, &o->gapo_del_cost };
        int t;
        for (t = 0; t < 2; ++t) {
            int c;
            if (r[t] > 0.0 && r[t] < 1.0) {
                c = (int)floor(U * log(r[t]) / lbar + 0.5);
                if (c < U) c = U;
                if (c > 8 * U) c = 8 * U;
            } else c = (int)floor(U * 11.0 / 3.0 + 0.5);
            *dst[t] = c;
        }
    }
    o->gape_cost = (int)floor(U * 4.0 / 3.0 + 0.5);
}

/* profile file formats: ErrorProfiling.java:504-531 (4 lines x 4 tab-separated
 * doubles, trailing tab) and :545-591 (one line "ins\tdel", no newline). */
int orc_read_profile_files(const char *ep, const char *ip, double P[16], double *ins, double *del)
{
    FILE *f = fopen(ep, "r");
    int i;
    if (!f) { FAIL("cannot open error profile %s", ep); return -1; }
    for (i = 0; i < 16; ++i) {
        char tok[64];
        if (fscanf(f, "%63s", tok) != 1) { fclose(f); FAIL("error profile %s: need 16 values", ep); return -1; }
        P[i] = strtod(tok, NULL); /* accepts NaN and 1.0E-4 */
    }
    fclose(f);
    *ins = *del = 0.0;
    if (ip && ip[0] && strcmp(ip, "null") != 0) {
        char t1[64], t2[64];
        f = fopen(ip, "r");
        if (!f) { FAIL("cannot open indel profile %s", ip); return -1; }
        if (fscanf(f, "%63s %63s", t1, t2) == 2) { *ins = strtod(t1, NULL); *del = strtod(t2, NULL); }
        fclose(f);
    }
    return 0;
}

/* ---------------------------------------------------------------- index -- */
typedef struct { char *name, *anno; int64_t offset; int32_t len, n_ambs; } ann_t;
typedef struct { int64_t offset; int32_t len; char amb; } hole_t;

#define OCC_SYMS 128            /* oracle's own block: 4 x u64 counts + 4 x u64 symbols */
struct orc_index {
    int64_t  l_pac;
    int      n_seqs, n_holes;
    ann_t   *anns; hole_t *holes;
    uint8_t *pac;                /* 2 bit, upstream _set_pac convention */
    uint64_t seq_len, primary, L2[5];
    uint64_t n_blocks; uint64_t *blk; /* per block: cnt[4], sym[4] (32 symbols per word, symbol j at bits 2j) */
    int      sa_intv; uint64_t n_sa; uint64_t *sa;
};

static const uint8_t nt4(int c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                 case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
#define PAC_SET(p, l, c) ((p)[(l) >> 2] |= (uint8_t)((c) << ((~(l) & 3) << 1)))
#define PAC_GET(p, l)    ((p)[(l) >> 2] >> ((~(l) & 3) << 1) & 3)

static char *read_file(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    char *buf; long n;
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    buf = (char *)malloc((size_t)n + 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); return NULL; }
    fclose(f); buf[n] = 0; *len = (size_t)n;
    return buf;
}

/* FASTA -> contig table, holes, forward pac (upstream bns_fasta2bntseq/add1:
 * non-ACGT becomes lrand48()&3 under srand48(11); a hole is a maximal run of
 * one identical non-ACGT character). */
static int load_fasta(orc_index_t *ix, const char *path)
{
    size_t n, i = 0;
    char *buf = read_file(path, &n);
    orc_rng_t rng;
    int64_t cap_pac = 1 << 20;
    int m_seqs = 0, m_holes = 0;
    if (!buf) { FAIL("cannot read %s", path); return -1; }
    orc_srand48(&rng, 11);
    ix->pac = (uint8_t *)calloc((size_t)cap_pac / 4 + 1, 1);
    while (i < n) {
        ann_t *a; hole_t *q = NULL; int lasts = 0; size_t s, e; int32_t pos = 0;
        while (i < n && buf[i] != '>') ++i;
        if (i >= n) break;
        ++i; s = i;
        while (i < n && !isspace((unsigned char)buf[i])) ++i;
        if (ix->n_seqs == m_seqs) { m_seqs = m_seqs ? m_seqs * 2 : 8; ix->anns = (ann_t *)realloc(ix->anns, sizeof(ann_t) * m_seqs); }
        a = &ix->anns[ix->n_seqs];
        a->name = strndup(buf + s, i - s);
        e = i; while (e < n && buf[e] != '\n') ++e;
        { size_t cs = i; while (cs < e && isspace((unsigned char)buf[cs])) ++cs;
          size_t ce = e; while (ce > cs && isspace((unsigned char)buf[ce - 1])) --ce;
          a->anno = ce > cs ? strndup(buf + cs, ce - cs) : strdup("(null)"); }
        i = e;
        a->offset = ix->n_seqs == 0 ? 0 : (a - 1)->offset + (a - 1)->len;
        a->n_ambs = 0;
        for (; i < n && buf[i] != '>'; ++i) {
            int ch = (unsigned char)buf[i], c;
            if (!isgraph(ch)) continue;
            c = nt4(ch);
            if (c >= 4) {
                if (lasts == ch) ++q->len;
                else {
                    if (ix->n_holes == m_holes) { m_holes = m_holes ? m_holes * 2 : 8; ix->holes = (hole_t *)realloc(ix->holes, sizeof(hole_t) * m_holes); }
                    q = &ix->holes[ix->n_holes++];
                    q->len = 1; q->offset = a->offset + pos; q->amb = (char)ch;
                    ++a->n_ambs;
                }
                c = (int)(orc_lrand48(&rng) & 3);
            }
            lasts = ch;
            if (ix->l_pac == cap_pac) {
                ix->pac = (uint8_t *)realloc(ix->pac, (size_t)cap_pac / 2 + 1);
                memset(ix->pac + cap_pac / 4, 0, (size_t)cap_pac / 4 + 1);
                cap_pac *= 2;
            }
            PAC_SET(ix->pac, ix->l_pac, c);
            ++ix->l_pac; ++pos;
        }
        a->len = pos;
        ++ix->n_seqs;
    }
    free(buf);
    if (ix->n_seqs == 0) { FAIL("no sequences in %s", path); return -1; }
    return 0;
}

/* LSD radix sort of (key,val) pairs, 16-bit digits */
static void rsort_kv(uint64_t *key, uint32_t *val, uint64_t n, int key_bits)
{
    uint64_t *k2 = (uint64_t *)malloc(n * 8); uint32_t *v2 = (uint32_t *)malloc(n * 4);
    size_t *cnt = (size_t *)malloc(65536 * sizeof(size_t));
    int sh;
    for (sh = 0; sh < key_bits; sh += 16) {
        uint64_t i; size_t s = 0, t; int d;
        memset(cnt, 0, 65536 * sizeof(size_t));
        for (i = 0; i < n; ++i) ++cnt[(key[i] >> sh) & 0xFFFF];
        for (d = 0; d < 65536; ++d) { t = cnt[d]; cnt[d] = s; s += t; }
        for (i = 0; i < n; ++i) { size_t p = cnt[(key[i] >> sh) & 0xFFFF]++; k2[p] = key[i]; v2[p] = val[i]; }
        { uint64_t *tk = key; key = k2; k2 = tk; } { uint32_t *tv = val; val = v2; v2 = tv; }
    }
    /* key_bits is 32 or 64 here: an even number of passes, so the result is back in the caller's arrays */
    free(k2); free(v2); free(cnt);
}

/* suffix array of T$ (n+1 suffixes) by prefix doubling; sa_out[0] == n */
static uint32_t *build_sa(const uint8_t *T, uint64_t n)
{
    uint64_t N = n + 1, i, h, ngroups;
    uint64_t *key = (uint64_t *)malloc(N * 8);
    uint32_t *sa = (uint32_t *)malloc(N * 4), *rank = (uint32_t *)malloc(N * 4);
    for (i = 0; i < N; ++i) {            /* 13 base-5 digits: symbol+1, 0 past the end */
        uint64_t k = 0; int j;
        for (j = 0; j < 13; ++j) k = k * 5 + (i + j < n ? (uint64_t)T[i + j] + 1 : 0);
        key[i] = k; sa[i] = (uint32_t)i;
    }
    rsort_kv(key, sa, N, 32);
    for (h = 13;; h *= 2) {
        uint32_t r = 0;
        ngroups = 0;
        for (i = 0; i < N; ++i) {
            if (i == 0 || key[i] != key[i - 1]) { r = (uint32_t)i; ++ngroups; }
            rank[sa[i]] = r;
        }
        if (ngroups == N) break;
        for (i = 0; i < N; ++i) {
            uint64_t p = sa[i];
            key[i] = ((uint64_t)rank[p] << 32) | (p + h < N ? rank[p + h] : 0);
        }
        rsort_kv(key, sa, N, 64);
    }
    free(key); free(rank);
    return sa;
}

static void build_occ(orc_index_t *ix, const uint8_t *B /* n symbols */)
{
    uint64_t n = ix->seq_len, nb = n / OCC_SYMS + 1, b, i, c[4] = {0, 0, 0, 0};
    ix->n_blocks = nb;
    ix->blk = (uint64_t *)calloc(nb * 8, 8);
    for (b = 0; b < nb; ++b) {
        uint64_t *q = ix->blk + b * 8;
        q[0] = c[0]; q[1] = c[1]; q[2] = c[2]; q[3] = c[3];
        for (i = b * OCC_SYMS; i < (b + 1) * OCC_SYMS && i < n; ++i) {
            uint64_t j = i - b * OCC_SYMS;
            q[4 + (j >> 5)] |= (uint64_t)B[i] << ((j & 31) << 1);
            ++c[B[i]];
        }
    }
    ix->L2[0] = 0;
    for (i = 0; i < 4; ++i) ix->L2[i + 1] = ix->L2[i] + c[i];
}

static __thread orc_stats_t g_st;
static orc_stats_t g_st_total;
static int g_blk_syms = 192;
void orc_set_block_syms(int s) { g_blk_syms = s; }
void orc_stats_reset(void) { memset(&g_st_total, 0, sizeof g_st_total); memset(&g_st, 0, sizeof g_st); }
static void stats_flush(void)
{
#pragma omp critical(orc_stats)
    {
        g_st_total.occ_pairs += g_st.occ_pairs; g_st_total.occ_same_blk += g_st.occ_same_blk;
        g_st_total.nodes += g_st.nodes; g_st_total.pushes += g_st.pushes; g_st_total.lf_steps += g_st.lf_steps;
        g_st_total.top2_breaks += g_st.top2_breaks; g_st_total.max_entries_stops += g_st.max_entries_stops;
        if (g_st.max_stack > g_st_total.max_stack) g_st_total.max_stack = g_st.max_stack;
    }
    memset(&g_st, 0, sizeof g_st);
}
void orc_stats_get(orc_stats_t *s) { stats_flush(); *s = g_st_total; }

static inline int popc_sym(uint64_t w, int c, int nsym) /* count symbol c among the low nsym symbols of w */
{
    uint64_t x = w ^ (0x5555555555555555ULL * (uint64_t)(3 - c)); /* positions equal to c become 11 */
    x = x & (x >> 1) & 0x5555555555555555ULL;
    if (nsym < 32) x &= (1ULL << (2 * nsym)) - 1;
    return __builtin_popcountll(x);
}

/* Occ(k,c) = #{rows r <= k of the BW matrix of T$ whose last column is c}; the
 * '$' row (primary) is not stored (upstream bwt_occ, bwt.c). */
uint64_t orc_occ(const orc_index_t *ix, int64_t k, int c)
{
    uint64_t kk, cnt; const uint64_t *q; int w, rem;
    if (k < 0) return 0;
    if ((uint64_t)k >= ix->seq_len) return ix->L2[c + 1] - ix->L2[c];
    kk = (uint64_t)k - ((uint64_t)k >= ix->primary); /* primary >= 1: row 0 is the '$' suffix */
    q = ix->blk + (kk / OCC_SYMS) * 8;
    cnt = q[c];
    rem = (int)(kk % OCC_SYMS) + 1; /* symbols 0..rem-1 of this block */
    for (w = 0; rem > 0; ++w, rem -= 32) cnt += popc_sym(q[4 + w], c, rem < 32 ? rem : 32);
    return cnt;
}

static inline void occ_pair_stat(const orc_index_t *ix, int64_t km1, int64_t l)
{
    ++g_st.occ_pairs;
    if (km1 >= 0 && (uint64_t)l < ix->seq_len) {
        uint64_t a = (uint64_t)km1 - ((uint64_t)km1 >= ix->primary), b = (uint64_t)l - ((uint64_t)l >= ix->primary);
        if (a / (uint64_t)g_blk_syms == b / (uint64_t)g_blk_syms) ++g_st.occ_same_blk;
    }
}

static inline int bwt_sym(const orc_index_t *ix, uint64_t i) /* stored symbol i */
{
    const uint64_t *q = ix->blk + (i / OCC_SYMS) * 8;
    uint64_t j = i % OCC_SYMS;
    return (int)(q[4 + (j >> 5)] >> ((j & 31) << 1) & 3);
}

/* upstream bwt_invPsi + bwt_sa (bwt.c): walk LF until a sampled row */
uint64_t orc_sa(const orc_index_t *ix, uint64_t k)
{
    uint64_t steps = 0, mask = (uint64_t)ix->sa_intv - 1;
    while (k & mask) {
        ++steps; ++g_st.lf_steps;
        if (k == ix->primary) k = 0;
        else {
            int c = bwt_sym(ix, k - (k > ix->primary));
            k = ix->L2[c] + orc_occ(ix, (int64_t)k, c);
        }
    }
    return steps + ix->sa[k / ix->sa_intv]; /* sa[0] == (uint64_t)-1, as upstream bwt_cal_sa */
}

static void finish_index_from_sa(orc_index_t *ix, const uint8_t *T, const uint32_t *SA)
{
    uint64_t n = ix->seq_len, i, j = 0;
    uint8_t *B = (uint8_t *)malloc(n ? n : 1);
    for (i = 0; i <= n; ++i) {
        if (SA[i] == 0) { ix->primary = i; continue; }
        B[j++] = T[SA[i] - 1];
    }
    build_occ(ix, B);
    free(B);
    ix->sa_intv = 32;
    ix->n_sa = (n + 32) / 32;
    ix->sa = (uint64_t *)malloc(ix->n_sa * 8);
    for (i = 0; i < ix->n_sa; ++i) ix->sa[i] = SA[i * 32];
    ix->sa[0] = (uint64_t)-1;
}

orc_index_t *orc_index_from_fasta(const char *fa)
{
    orc_index_t *ix = (orc_index_t *)calloc(1, sizeof *ix);
    uint8_t *T; uint32_t *SA; int64_t i;
    if (load_fasta(ix, fa) < 0) { orc_index_free(ix); return NULL; }
    if ((uint64_t)ix->l_pac * 2 + 1 >= 0xFFFFFFFFULL) { FAIL("oracle SA builder is 32-bit; genome too large"); orc_index_free(ix); return NULL; }
    ix->seq_len = (uint64_t)ix->l_pac * 2;
    T = (uint8_t *)malloc(ix->seq_len + 1);
    for (i = 0; i < ix->l_pac; ++i) T[i] = PAC_GET(ix->pac, i);
    for (i = 0; i < ix->l_pac; ++i) T[ix->l_pac + i] = 3 - T[ix->l_pac - 1 - i]; /* forward + reverse complement, one BWT */
    SA = build_sa(T, ix->seq_len);
    finish_index_from_sa(ix, T, SA);
    free(SA); free(T);
    return ix;
}

orc_index_t *orc_index_from_parts(const char *fa, const uint8_t *bwt_syms, uint64_t n, uint64_t primary,
                                  const uint64_t *sa_samples, int sa_intv)
{
    orc_index_t *ix = (orc_index_t *)calloc(1, sizeof *ix);
    if (load_fasta(ix, fa) < 0) { orc_index_free(ix); return NULL; }
    if ((uint64_t)ix->l_pac * 2 != n) { FAIL("bwt length %llu != 2*l_pac", (unsigned long long)n); orc_index_free(ix); return NULL; }
    ix->seq_len = n; ix->primary = primary;
    build_occ(ix, bwt_syms);
    ix->sa_intv = sa_intv; ix->n_sa = (n + sa_intv) / sa_intv;
    ix->sa = (uint64_t *)malloc(ix->n_sa * 8);
    memcpy(ix->sa, sa_samples, ix->n_sa * 8);
    ix->sa[0] = (uint64_t)-1;
    return ix;
}

void orc_index_free(orc_index_t *ix)
{
    int i;
    if (!ix) return;
    for (i = 0; i < ix->n_seqs; ++i) { free(ix->anns[i].name); free(ix->anns[i].anno); }
    free(ix->anns); free(ix->holes); free(ix->pac); free(ix->blk); free(ix->sa); free(ix);
}
uint64_t orc_index_seq_len(const orc_index_t *ix) { return ix->seq_len; }
uint64_t orc_index_l_pac(const orc_index_t *ix) { return (uint64_t)ix->l_pac; }
uint64_t orc_index_primary(const orc_index_t *ix) { return ix->primary; }
void orc_index_L2(const orc_index_t *ix, uint64_t out[5]) { memcpy(out, ix->L2, 40); }
int orc_index_n_seqs(const orc_index_t *ix) { return ix->n_seqs; }
int orc_index_n_holes(const orc_index_t *ix) { return ix->n_holes; }
const uint8_t *orc_index_pac(const orc_index_t *ix) { return ix->pac; }
void orc_index_bwt_syms(const orc_index_t *ix, uint8_t *out) { uint64_t i; for (i = 0; i < ix->seq_len; ++i) out[i] = (uint8_t)bwt_sym(ix, i); }
uint64_t orc_index_n_sa(const orc_index_t *ix) { return ix->n_sa; }
void orc_index_sa_samples(const orc_index_t *ix, uint64_t *out) { memcpy(out, ix->sa, ix->n_sa * 8); }

/* ------------------------------------------------------ search: widths --- */
static inline void occ2(const orc_index_t *ix, uint64_t k, uint64_t l, int c, uint64_t *ok, uint64_t *ol)
{
    occ_pair_stat(ix, (int64_t)k - 1, (int64_t)l);
    *ok = orc_occ(ix, (int64_t)k - 1, c);
    *ol = orc_occ(ix, (int64_t)l, c);
}

/* upstream bwt_cal_width (bwtaln.c).  str = read reversed (not complemented):
 * width[i] = (interval size, number of restarts) for the read's last i+1
 * bases; by strand symmetry of T this bounds the differences needed for
 * search positions 0..i. */
int orc_cal_width(const orc_index_t *ix, int len, const uint8_t *str, orc_width_t *width)
{
    uint64_t k = 0, l = ix->seq_len, ok, ol;
    int i, bid = 0;
    for (i = 0; i < len; ++i) {
        int c = str[i];
        if (c < 4) {
            occ2(ix, k, l, c, &ok, &ol);
            k = ix->L2[c] + ok + 1;
            l = ix->L2[c] + ol;
        }
        if (k > l || c > 3) { k = 0; l = ix->seq_len; ++bid; }
        width[i].w = l - k + 1;
        width[i].bid = bid;
    }
    width[len].w = 0;
    width[len].bid = ++bid;
    return bid;
}

/* ---------------------------------------------- search: bounded backtrack -- */
enum { ST_M = 0, ST_I = 1, ST_D = 2 };
typedef struct {
    uint64_t k, l;
    int32_t i, score, units;
    uint8_t n_mm, n_gapo, n_gape, n_ins, n_del, state;
    int16_t last_diff_pos;
} entry_t;

typedef struct { entry_t *a; int n, m; } bucket_t;
typedef struct { bucket_t *b; int n_buckets, best, n_entries; } heap_t;

typedef struct {
    int u_mm[5][4], s_mm[5][4];
    int u_gapo_ins, s_gapo_ins, u_gapo_del, s_gapo_del, u_gape, s_gape;
    int s_stop, u_tight, c_min, max_units, n_buckets;
} model_t;

/* budget for a read of this length: upstream bwa_cal_sa_reg_gap (fnr>0 => per
 * length Poisson table with 2% error) */
static int budget_diffs(const orc_opt_t *o, int len)
{
    if (o->profile) return o->x_avg_mm >= 0 ? o->x_avg_mm : orc_cal_maxdiff(len, 0.02, 0.04);
    return o->fnr > 0.0 ? orc_cal_maxdiff(len, 0.02, o->fnr) : o->max_diff;
}

static void make_model(const orc_opt_t *o, int len, model_t *m)
{
    int s, c, max_cost = 0;
    memset(m, 0, sizeof *m);
    if (!o->profile) {
        int md = budget_diffs(o, len), mg = o->max_gapo < md ? o->max_gapo : md; /* upstream clamps max_gapo to max_diff */
        for (s = 0; s < 5; ++s) for (c = 0; c < 4; ++c) { int mm = (s != c); m->u_mm[s][c] = mm; m->s_mm[s][c] = mm * o->s_mm; }
        m->u_gapo_ins = m->u_gapo_del = 1; m->s_gapo_ins = m->s_gapo_del = o->s_gapo;
        m->u_gape = o->mode_gape ? 1 : 0; m->s_gape = o->s_gape;
        m->s_stop = o->s_mm; m->u_tight = 1; m->c_min = 1; m->max_units = md;
        m->n_buckets = (md + 1) * o->s_mm + (mg + 1) * o->s_gapo + (o->max_gape + 1) * o->s_gape; /* upstream gap_init_stack */
        if (m->n_buckets < 1) m->n_buckets = 1;
    } else {
        int U = o->unit;
        m->c_min = 1 << 30;
        for (s = 0; s < 5; ++s) for (c = 0; c < 4; ++c) {
            /* search works on the reverse-complemented read against T: ref base (read
             * orientation) = 3-c, read base = 3-s */
            int cost = s == 4 ? o->n_cost : (s == c ? 0 : o->sub_cost[(3 - c) * 4 + (3 - s)]);
            m->u_mm[s][c] = m->s_mm[s][c] = cost;
            if (cost > 0 && cost < m->c_min) m->c_min = cost;
            if (cost > max_cost) max_cost = cost;
        }
        m->u_gapo_ins = m->s_gapo_ins = o->gapo_ins_cost;
        m->u_gapo_del = m->s_gapo_del = o->gapo_del_cost;
        m->u_gape = m->s_gape = o->gape_cost;
        if (o->gapo_ins_cost < m->c_min) m->c_min = o->gapo_ins_cost;
        if (o->gapo_del_cost < m->c_min) m->c_min = o->gapo_del_cost;
        if (o->gape_cost < m->c_min) m->c_min = o->gape_cost;
        if (o->gapo_ins_cost > max_cost) max_cost = o->gapo_ins_cost;
        if (o->gapo_del_cost > max_cost) max_cost = o->gapo_del_cost;
        if (o->gape_cost > max_cost) max_cost = o->gape_cost;
        if (m->c_min < 1) m->c_min = 1;
        m->s_stop = U; m->u_tight = U;
        m->max_units = budget_diffs(o, len) * U;
        m->n_buckets = m->max_units + max_cost + 1;
    }
}

static void heap_push(heap_t *h, const entry_t *e)
{
    bucket_t *q = &h->b[e->score];
    if (q->n == q->m) { q->m = q->m ? q->m * 2 : 4; q->a = (entry_t *)realloc(q->a, sizeof(entry_t) * q->m); }
    q->a[q->n++] = *e;
    ++h->n_entries;
    if (h->best > e->score) h->best = e->score;
    ++g_st.pushes;
    if ((uint64_t)h->n_entries > g_st.max_stack) g_st.max_stack = h->n_entries;
}
/* upstream gap_pop (bwtgap.c): newest entry of the lowest non-empty score bucket */
static void heap_pop(heap_t *h, entry_t *e)
{
    bucket_t *q = &h->b[h->best];
    *e = q->a[--q->n];
    --h->n_entries;
    if (q->n == 0 && h->n_entries) {
        int i;
        for (i = h->best + 1; i < h->n_buckets; ++i) if (h->b[i].n) break;
        h->best = i;
    } else if (h->n_entries == 0) h->best = h->n_buckets;
}

static int match_exact_alt(const orc_index_t *ix, int len, const uint8_t *str, uint64_t *k0, uint64_t *l0)
{
    uint64_t k = *k0, l = *l0, ok, ol;
    int i;
    for (i = len - 1; i >= 0; --i) {
        int c = str[i];
        if (c > 3) return 0;
        occ2(ix, k, l, c, &ok, &ol);
        k = ix->L2[c] + ok + 1;
        l = ix->L2[c] + ol;
        if (k > l) return 0;
    }
    *k0 = k; *l0 = l;
    return 1;
}

/* upstream gap_shadow (bwtgap.c) */
static void shadow(uint64_t x, uint64_t max, int last_diff_pos, orc_width_t *w)
{
    int i, j;
    for (i = j = 0; i < last_diff_pos; ++i) {
        if (w[i].w > x) w[i].w -= x;
        else if (w[i].w == x) { w[i].bid = 1; w[i].w = max - (uint64_t)(++j); }
    }
}

#define PUSH(I, K, L, MM, GO, GE, NI, ND, ST, ISDIFF, SC, UN) do { \
        if ((UN) <= max_units) { entry_t c_; c_.i = (I); c_.k = (K); c_.l = (L); c_.n_mm = (uint8_t)(MM); c_.n_gapo = (uint8_t)(GO); \
        c_.n_gape = (uint8_t)(GE); c_.n_ins = (uint8_t)(NI); c_.n_del = (uint8_t)(ND); c_.state = (ST); \
        c_.last_diff_pos = (int16_t)((ISDIFF) ? (I) : 0); c_.score = (SC); c_.units = (UN); heap_push(&h, &c_); } } while (0)

/* upstream bwt_match_gap (bwtgap.c); seq = reverse complement of the read.
 * Differences from the stock routine, all no-ops when profile==0: edit costs
 * come from model_t, the difference budget is held in units, and a child whose
 * units exceed the budget is not pushed. */
static int match_gap(const orc_index_t *ix, int len, const uint8_t *seq, orc_width_t *width, orc_width_t *seed_width,
                     const orc_opt_t *o, const model_t *md, orc_aln_t *out, int cap)
{
    heap_t h;
    int best_score = 1 << 29, max_units = md->max_units, n_aln = 0, j, nN = 0, nNu = 0;
    int max_gapo = o->max_gapo;
    uint64_t best_cnt = 0;
    int seed_len = o->seed_len < len ? o->seed_len : 0x7fffffff;
    if (!o->profile && max_units < max_gapo) max_gapo = max_units;
    for (j = 0; j < len; ++j) if (seq[j] > 3) { ++nN; nNu += md->u_mm[4][0]; }
    if (nNu > max_units) return 0;
    (void)nN;
    h.n_buckets = md->n_buckets; h.best = h.n_buckets; h.n_entries = 0;
    h.b = (bucket_t *)calloc((size_t)h.n_buckets, sizeof(bucket_t));
    { entry_t r; memset(&r, 0, sizeof r); r.i = len; r.k = 0; r.l = ix->seq_len; heap_push(&h, &r); }
    while (h.n_entries) {
        entry_t e; int i, m, m_seed = 0, hit, allow_diff, allow_M, tmp, rem;
        uint64_t k, l, ck[4], cl[4], occ;
        if (h.n_entries > o->max_entries) { ++g_st.max_entries_stops; break; }
        heap_pop(&h, &e);
        k = e.k; l = e.l; i = e.i;
        if (e.score > best_score + md->s_stop) break;
        rem = max_units - e.units;
        if (rem < 0) continue;
        m = rem / md->c_min;
        /* seed budget in units as well (stock: max_seed_diff - ndiff) */
        if (seed_width) m_seed = (o->max_seed_diff * md->u_tight - e.units) / md->c_min;
        if (i > 0 && m < width[i - 1].bid) continue;
        hit = 0;
        if (i == 0) hit = 1;
        else if (m == 0 && (e.state == ST_M || o->mode_gape || e.n_gape == o->max_gape)) {
            if (match_exact_alt(ix, i, seq, &k, &l)) hit = 1; else continue;
        }
        if (hit) {
            int do_add = 1;
            if (n_aln == 0) {
                best_score = e.score;
                max_units = e.units + md->u_tight > md->max_units ? md->max_units : e.units + md->u_tight;
            }
            if (e.score == best_score) best_cnt += l - k + 1;
            else if (best_cnt > (uint64_t)o->max_top2) { ++g_st.top2_breaks; break; }
            if (e.n_gapo) {
                int t, lim = n_aln < cap ? n_aln : cap;
                for (t = 0; t < lim; ++t) if (out[t].k == k && out[t].l == l) break;
                if (t < lim) do_add = 0;
            }
            if (do_add) {
                shadow(l - k + 1, ix->seq_len, e.last_diff_pos, width);
                if (n_aln < cap) {
                    orc_aln_t *p = &out[n_aln];
                    p->k = k; p->l = l; p->n_mm = e.n_mm; p->n_gapo = e.n_gapo; p->n_gape = e.n_gape;
                    p->n_ins = e.n_ins; p->n_del = e.n_del; p->score = e.score; p->units = e.units; p->_pad = 0;
                }
                ++n_aln;
            }
            continue;
        }
        --i;
        ++g_st.nodes;
        occ_pair_stat(ix, (int64_t)k - 1, (int64_t)l);
        for (j = 0; j < 4; ++j) { ck[j] = orc_occ(ix, (int64_t)k - 1, j); cl[j] = orc_occ(ix, (int64_t)l, j); }
        occ = l - k + 1;
        allow_diff = allow_M = 1;
        if (i > 0) {
            int ii = i - (len - seed_len);
            if (width[i - 1].bid > m - 1) allow_diff = 0;
            else if (width[i - 1].bid == m - 1 && width[i].bid == m - 1 && width[i - 1].w == width[i].w) allow_M = 0;
            if (seed_width && ii > 0) {
                if (seed_width[ii - 1].bid > m_seed - 1) allow_diff = 0;
                else if (seed_width[ii - 1].bid == m_seed - 1 && seed_width[ii].bid == m_seed - 1
                         && seed_width[ii - 1].w == seed_width[ii].w) allow_M = 0;
            }
        }
        tmp = e.n_gapo + e.n_gape;
        if (allow_diff && i >= o->indel_end_skip + tmp && len - i >= o->indel_end_skip + tmp) {
            if (e.state == ST_M) {
                if (e.n_gapo < max_gapo) {
                    PUSH(i, k, l, e.n_mm, e.n_gapo + 1, e.n_gape, e.n_ins + 1, e.n_del, ST_I, 1,
                         e.score + md->s_gapo_ins, e.units + md->u_gapo_ins);
                    for (j = 0; j < 4; ++j) {
                        uint64_t nk = ix->L2[j] + ck[j] + 1, nl = ix->L2[j] + cl[j];
                        if (nk <= nl) PUSH(i + 1, nk, nl, e.n_mm, e.n_gapo + 1, e.n_gape, e.n_ins, e.n_del + 1, ST_D, 1,
                                           e.score + md->s_gapo_del, e.units + md->u_gapo_del);
                    }
                }
            } else if (e.state == ST_I) {
                if (e.n_gape < o->max_gape)
                    PUSH(i, k, l, e.n_mm, e.n_gapo, e.n_gape + 1, e.n_ins + 1, e.n_del, ST_I, 1,
                         e.score + md->s_gape, e.units + md->u_gape);
            } else {
                if (e.n_gape < o->max_gape) {
                    if ((e.n_gape + e.n_gapo) * md->u_tight < max_units || occ < (uint64_t)o->max_del_occ) {
                        for (j = 0; j < 4; ++j) {
                            uint64_t nk = ix->L2[j] + ck[j] + 1, nl = ix->L2[j] + cl[j];
                            if (nk <= nl) PUSH(i + 1, nk, nl, e.n_mm, e.n_gapo, e.n_gape + 1, e.n_ins, e.n_del + 1, ST_D, 1,
                                               e.score + md->s_gape, e.units + md->u_gape);
                        }
                    }
                }
            }
        }
        if (allow_diff && allow_M) {
            for (j = 1; j <= 4; ++j) {
                int c = (seq[i] + j) & 3, is_mm = (j != 4 || seq[i] > 3);
                uint64_t nk = ix->L2[c] + ck[c] + 1, nl = ix->L2[c] + cl[c];
                if (nk <= nl) PUSH(i, nk, nl, e.n_mm + is_mm, e.n_gapo, e.n_gape, e.n_ins, e.n_del, ST_M, is_mm,
                                   e.score + (is_mm ? md->s_mm[seq[i]][c] : 0), e.units + (is_mm ? md->u_mm[seq[i]][c] : 0));
            }
        } else if (seq[i] < 4) {
            int c = seq[i] & 3;
            uint64_t nk = ix->L2[c] + ck[c] + 1, nl = ix->L2[c] + cl[c];
            if (nk <= nl) PUSH(i, nk, nl, e.n_mm, e.n_gapo, e.n_gape, e.n_ins, e.n_del, ST_M, 0, e.score, e.units);
        }
    }
    for (j = 0; j < h.n_buckets; ++j) free(h.b[j].a);
    free(h.b);
    return n_aln;
}

/* upstream bwa_cal_sa_reg_gap body for one read (bwtaln.c).  read = codes in
 * read orientation. */
int orc_aln_one(const orc_index_t *ix, const orc_opt_t *o, int len, const uint8_t *read,
                orc_aln_t *out, int cap, orc_width_t *width_out, orc_width_t *seed_width_out)
{
    uint8_t *rev = (uint8_t *)malloc((size_t)len + 1);
    orc_width_t *w = (orc_width_t *)calloc((size_t)len + 1, sizeof *w), *sw = NULL;
    model_t md;
    int i, n;
    for (i = 0; i < len; ++i) rev[i] = read[len - 1 - i];
    orc_cal_width(ix, len, rev, w);
    if (len > o->seed_len) {
        sw = (orc_width_t *)calloc((size_t)o->seed_len + 1, sizeof *sw);
        orc_cal_width(ix, o->seed_len, rev + (len - o->seed_len), sw);
    }
    if (width_out) memcpy(width_out, w, sizeof(*w) * ((size_t)len + 1));
    if (seed_width_out && sw) memcpy(seed_width_out, sw, sizeof(*sw) * ((size_t)o->seed_len + 1));
    for (i = 0; i < len; ++i) rev[i] = rev[i] > 3 ? 4 : 3 - rev[i];
    make_model(o, len, &md);
    n = match_gap(ix, len, rev, w, sw, o, &md, out, cap);
    free(rev); free(w); free(sw);
    return n;
}

/* ------------------------------------------------- banded global alignment -- */
/* upstream ksw_global (ksw.c) as called by bwa_refine_gapped_core (bwase.c):
 * match 1, mismatch -3, N -1, gap open 5, extend 1.  cigar: len<<4|op, op 0 M,
 * 1 I, 2 D.  Target in the outer loop; z keeps 2 bits of H direction and one
 * continuation bit each for E and F. */
int orc_ksw_global(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w,
                   uint32_t *cigar, int cap)
{
    const int NEG = -0x40000000, gapo = 5, gape = 1, gapoe = gapo + gape;
    int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1, i, j, k, n_cigar = 0, which = 0;
    uint8_t *z = (uint8_t *)malloc((size_t)n_col * tlen + 1);
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * (qlen + 1)), *E = (int32_t *)malloc(sizeof(int32_t) * (qlen + 1));
    H[0] = 0; E[0] = NEG;
    for (j = 1; j <= qlen && j <= w; ++j) { H[j] = -(gapo + gape * j); E[j] = NEG; }
    for (; j <= qlen; ++j) H[j] = E[j] = NEG;
    for (i = 0; i < tlen; ++i) {
        int32_t f = NEG, h1, beg, end;
        uint8_t *zi = z + (size_t)i * n_col;
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(gapo + gape * (i + 1)) : NEG;
        for (j = beg; j < end; ++j) {
            int32_t h = H[j], e = E[j], sc;
            uint8_t d;
            H[j] = h1;
            sc = (query[j] > 3 || target[i] > 3) ? -1 : (query[j] == target[i] ? 1 : -3);
            h += sc;
            d = h >= e ? 0 : 1; h = h >= e ? h : e;
            d = h >= f ? d : 2; h = h >= f ? h : f;
            h1 = h;
            h -= gapoe;
            e -= gape;
            d |= e > h ? 1 << 2 : 0; e = e > h ? e : h;
            E[j] = e;
            f -= gape;
            d |= f > h ? 2 << 4 : 0; f = f > h ? f : h;
            zi[j - beg] = d;
        }
        H[end] = h1; E[end] = NEG;
    }
    /* backtrack */
#define PUSHC(OP, LEN) do { if (n_cigar && (cigar[n_cigar - 1] & 0xf) == (uint32_t)(OP)) cigar[n_cigar - 1] += (uint32_t)(LEN) << 4; \
        else if (n_cigar < cap) cigar[n_cigar++] = (uint32_t)(LEN) << 4 | (OP); } while (0)
    i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    while (i >= 0 && k >= 0) {
        which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
        if (which == 0) { PUSHC(0, 1); --i; --k; }
        else if (which == 1) { PUSHC(2, 1); --i; }
        else { PUSHC(1, 1); --k; }
    }
    if (i >= 0) PUSHC(2, i + 1);
    if (k >= 0) PUSHC(1, k + 1);
    for (i = 0; i < n_cigar >> 1; ++i) { uint32_t t = cigar[i]; cigar[i] = cigar[n_cigar - 1 - i]; cigar[n_cigar - 1 - i] = t; }
    free(z); free(H); free(E);
    return n_cigar;
}

/* ---------------------------------------------------------- samse stage -- */
typedef struct { uint64_t pos; int32_t gap, mm, ref_shift, strand, n_cigar; uint32_t cigar[16]; } multi_t;
typedef struct {
    char *name; uint8_t *seq /* read orientation */, *rseq /* reverse complement */; char *qual; int len;
    int n_aln; orc_aln_t *aln;
    int type, strand, mapq, n_mm, n_gapo, n_gape, ref_shift, score, c1, c2, nm, n_cigar, n_multi;
    uint64_t sa, pos;
    uint32_t cigar[16];
    char *md;
    multi_t *multi;
} read_t;

/* upstream bwa_aln2seq_core (bwase.c), set_main=1 */
static void aln2seq(read_t *s, int n_multi, orc_rng_t *rng)
{
    int i, n_aln = s->n_aln, cnt = 0, best; /* cnt is an int upstream */
    const orc_aln_t *aln = s->aln;
    if (n_aln == 0) { s->type = 0; s->c1 = s->c2 = 0; return; }
    best = aln[0].score;
    for (i = 0; i < n_aln; ++i) {
        const orc_aln_t *p = aln + i;
        if (p->score > best) break;
        if (orc_drand48(rng) * (double)(p->l - p->k + 1 + (uint64_t)(int64_t)cnt) > (double)cnt) {
            s->n_mm = p->n_mm; s->n_gapo = p->n_gapo; s->n_gape = p->n_gape;
            s->ref_shift = p->n_del - p->n_ins; s->score = p->score;
            s->sa = p->k + (uint64_t)((double)(p->l - p->k + 1) * orc_drand48(rng));
        }
        cnt += (int)(p->l - p->k + 1);
    }
    s->c1 = cnt;
    for (; i < n_aln; ++i) cnt += (int)(aln[i].l - aln[i].k + 1);
    s->c2 = cnt - s->c1;
    s->type = s->c1 > 1 ? 2 : 1;
    if (n_multi) {
        int k, rest, z = 0, n_occ = 0;
        for (k = 0; k < n_aln; ++k) n_occ += (int)(aln[k].l - aln[k].k + 1);
        if (n_occ > n_multi + 1 || n_occ < 0) { s->multi = NULL; s->n_multi = 0; return; }
        rest = n_occ;
        s->multi = (multi_t *)calloc((size_t)rest + 1, sizeof(multi_t));
        for (k = 0; k < n_aln; ++k) {
            const orc_aln_t *q = aln + k; uint64_t l;
            for (l = q->k; l <= q->l; ++l) {
                s->multi[z].pos = l; s->multi[z].gap = q->n_gapo + q->n_gape;
                s->multi[z].ref_shift = q->n_del - q->n_ins; s->multi[z].mm = q->n_mm; ++z;
            }
        }
        s->n_multi = z;
    }
}

/* upstream bwa_sa2pos (bwase.c) */
static uint64_t sa2pos(const orc_index_t *ix, uint64_t sapos, int ref_len, int *strand)
{
    uint64_t pos_f = orc_sa(ix, sapos), l_pac = (uint64_t)ix->l_pac;
    int is_rev;
    *strand = 0; /* upstream leaves it unset on this path; fixed to 0 here */
    if (pos_f < l_pac && l_pac < pos_f + (uint64_t)ref_len) return (uint64_t)-1;
    is_rev = pos_f >= l_pac;
    if (is_rev) pos_f = (l_pac << 1) - 1 - pos_f;
    *strand = !is_rev;
    if (is_rev) pos_f = pos_f + 1 < (uint64_t)ref_len ? 0 : pos_f - (uint64_t)ref_len + 1;
    return pos_f;
}

/* upstream bwa_approx_mapQ (bwase.c); profile mode: budget test in units */
static int approx_mapq(const read_t *p, const orc_opt_t *o, int hit_units)
{
    int n, budget = budget_diffs(o, p->len);
    if (p->c1 == 0) return 23;
    if (p->c1 > 1) return 0;
    if (!o->profile) { if (p->n_mm == budget) return 25; }
    else if (budget * o->unit - hit_units < o->unit) return 25;
    if (p->c2 == 0) return 37;
    n = p->c2 >= 255 ? 255 : p->c2;
    return 23 < orc_mapq_logn(n) ? 0 : 23 - orc_mapq_logn(n);
}

/* upstream bwa_refine_gapped_core (bwase.c) */
static int refine_gapped(const orc_index_t *ix, int len, const uint8_t *seq, int ref_shift, uint64_t *_rb, uint32_t *cigar)
{
    int64_t rb = (int64_t)*_rb, re = rb + len + ref_shift, rlen, k;
    uint8_t *rseq; int n_cigar, w;
    if (re > ix->l_pac) re = ix->l_pac; /* upstream asserts re <= l_pac */
    if (rb >= re) return 0;
    rlen = re - rb;
    rseq = (uint8_t *)malloc((size_t)rlen);
    for (k = 0; k < rlen; ++k) rseq[k] = PAC_GET(ix->pac, rb + k);
    w = (int)(abs((int)rlen - len) * 1.5);
    n_cigar = orc_ksw_global(len, seq, (int)rlen, rseq, 50 > w ? 50 : w, cigar, 16);
    free(rseq);
    if (n_cigar <= 0) return 0;
    if ((cigar[n_cigar - 1] & 0xf) == 1) cigar[n_cigar - 1] = (cigar[n_cigar - 1] >> 4 << 4) | 3;
    if ((cigar[0] & 0xf) == 1) cigar[0] = (cigar[0] >> 4 << 4) | 3;
    if ((cigar[n_cigar - 1] & 0xf) == 2) --n_cigar;
    if (n_cigar > 0 && (cigar[0] & 0xf) == 2) {
        *_rb += cigar[0] >> 4;
        --n_cigar;
        memmove(cigar, cigar + 1, (size_t)n_cigar * 4);
    }
    return n_cigar;
}

typedef struct { char *s; size_t l, m; } str_t;
static void sputc(str_t *s, int c) { if (s->l + 2 > s->m) { s->m = s->m ? s->m * 2 : 64; s->s = (char *)realloc(s->s, s->m); } s->s[s->l++] = (char)c; s->s[s->l] = 0; }
static void sputs(str_t *s, const char *p) { while (*p) sputc(s, *p++); }
static void sputi(str_t *s, long v) { char b[32]; snprintf(b, sizeof b, "%ld", v); sputs(s, b); }

/* upstream bwa_cal_md1 (bwase.c) */
static char *cal_md(const orc_index_t *ix, int n_cigar, const uint32_t *cigar, int len, uint64_t pos, const uint8_t *seq, int *_nm)
{
    uint64_t x = pos, y = 0, l_pac = (uint64_t)ix->l_pac;
    int z, u = 0, c, nm = 0;
    str_t s = {0, 0, 0};
    if (n_cigar) {
        int k;
        for (k = 0; k < n_cigar; ++k) {
            int l = (int)(cigar[k] >> 4), op = (int)(cigar[k] & 0xf);
            if (op == 0) {
                for (z = 0; z < l && x + z < l_pac; ++z) {
                    c = PAC_GET(ix->pac, x + z);
                    if (seq[y + z] > 3 || c != seq[y + z]) { sputi(&s, u); sputc(&s, "ACGTN"[c]); ++nm; u = 0; }
                    else ++u;
                }
                x += l; y += l;
            } else if (op == 1 || op == 3) {
                y += l;
                if (op == 1) nm += l;
            } else if (op == 2) {
                sputi(&s, u); sputc(&s, '^');
                for (z = 0; z < l && x + z < l_pac; ++z) sputc(&s, "ACGT"[PAC_GET(ix->pac, x + z)]);
                u = 0; x += l; nm += l;
            }
        }
    } else {
        for (z = u = 0; z < len && x + z < l_pac; ++z) {
            c = PAC_GET(ix->pac, x + z);
            if (seq[y + z] > 3 || c != seq[y + z]) { sputi(&s, u); sputc(&s, "ACGTN"[c]); ++nm; u = 0; }
            else ++u;
        }
    }
    sputi(&s, u);
    *_nm = nm;
    return s.s;
}

static int pos2rid(const orc_index_t *ix, int64_t pos_f) /* upstream bns_pos2rid */
{
    int left = 0, mid = 0, right = ix->n_seqs;
    if (pos_f >= ix->l_pac) return -1;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= ix->anns[mid].offset) {
            if (mid == ix->n_seqs - 1) break;
            if (pos_f < ix->anns[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
static int cnt_ambi(const orc_index_t *ix, int64_t pos_f, int len, int *ref_id) /* upstream bns_cnt_ambi */
{
    int left = 0, right = ix->n_holes, nn = 0, mid;
    if (ref_id) *ref_id = pos2rid(ix, pos_f);
    while (left < right) {
        const hole_t *h;
        mid = (left + right) >> 1; h = &ix->holes[mid];
        if (pos_f >= h->offset + h->len) left = mid + 1;
        else if (pos_f + len <= h->offset) right = mid;
        else {
            if (pos_f >= h->offset) nn += h->offset + h->len < pos_f + len ? (int)(h->offset + h->len - pos_f) : len;
            else nn += h->offset + h->len < pos_f + len ? h->len : len - (int)(h->offset - pos_f);
            break;
        }
    }
    return nn;
}

static int64_t ref_span(int n_cigar, const uint32_t *cigar, int len)
{
    int j; int64_t x = 0;
    if (!n_cigar) return len;
    for (j = 0; j < n_cigar; ++j) { int op = (int)(cigar[j] & 0xf); if (op == 0 || op == 2) x += cigar[j] >> 4; }
    return x;
}

static void put_cigar(str_t *o, int n_cigar, const uint32_t *cigar, int len)
{
    int j;
    if (n_cigar) for (j = 0; j < n_cigar; ++j) { sputi(o, cigar[j] >> 4); sputc(o, "MIDS"[cigar[j] & 0xf]); }
    else { sputi(o, len); sputc(o, 'M'); }
}

/* upstream bwa_print_sam1 (bwase.c), single-end, no read group */
static void print_sam(const orc_index_t *ix, const orc_opt_t *o, read_t *p, str_t *out, orc_hit_t *hit)
{
    int j;
    if (hit) { memset(hit, 0, sizeof *hit); hit->pos = -1; hit->seqid = -1; }
    if (p->type != 0) {
        int seqid, nn, flag = 0, i;
        j = (int)ref_span(p->n_cigar, p->cigar, p->len);
        nn = cnt_ambi(ix, (int64_t)p->pos, j, &seqid);
        if ((int64_t)p->pos + j - ix->anns[seqid].offset > ix->anns[seqid].len) flag |= 4;
        if (p->strand) flag |= 16;
        sputs(out, p->name); sputc(out, '\t'); sputi(out, flag); sputc(out, '\t'); sputs(out, ix->anns[seqid].name); sputc(out, '\t');
        sputi(out, (long)((int64_t)p->pos - ix->anns[seqid].offset + 1)); sputc(out, '\t'); sputi(out, p->mapq); sputc(out, '\t');
        put_cigar(out, p->n_cigar, p->cigar, p->len);
        sputs(out, "\t*\t0\t0\t");
        if (!p->strand) for (i = 0; i < p->len; ++i) sputc(out, "ACGTN"[p->seq[i]]);
        else for (i = p->len - 1; i >= 0; --i) sputc(out, "TGCAN"[p->seq[i]]);
        sputc(out, '\t');
        if (p->qual) { if (!p->strand) sputs(out, p->qual); else for (i = p->len - 1; i >= 0; --i) sputc(out, p->qual[i]); }
        else sputc(out, '*');
        {
            char XT = "NURM"[p->type];
            if (nn > 10) XT = 'N';
            sputs(out, "\tXT:A:"); sputc(out, XT); sputs(out, "\tNM:i:"); sputi(out, p->nm);
            if (nn) { sputs(out, "\tXN:i:"); sputi(out, nn); }
            sputs(out, "\tX0:i:"); sputi(out, p->c1);
            if (p->c1 <= o->max_top2) { sputs(out, "\tX1:i:"); sputi(out, p->c2); }
            sputs(out, "\tXM:i:"); sputi(out, p->n_mm); sputs(out, "\tXO:i:"); sputi(out, p->n_gapo);
            sputs(out, "\tXG:i:"); sputi(out, p->n_gapo + p->n_gape);
            if (p->md) { sputs(out, "\tMD:Z:"); sputs(out, p->md); }
            if (p->n_multi) {
                sputs(out, "\tXA:Z:");
                for (i = 0; i < p->n_multi; ++i) {
                    multi_t *q = p->multi + i; int sid;
                    j = (int)ref_span(q->n_cigar, q->cigar, p->len);
                    cnt_ambi(ix, (int64_t)q->pos, j, &sid);
                    sputs(out, ix->anns[sid].name); sputc(out, ','); sputc(out, q->strand ? '-' : '+');
                    sputi(out, (long)((int64_t)q->pos - ix->anns[sid].offset + 1)); sputc(out, ',');
                    put_cigar(out, q->n_cigar, q->cigar, p->len);
                    sputc(out, ','); sputi(out, q->gap + q->mm); sputc(out, ';');
                }
            }
        }
        sputc(out, '\n');
        if (hit) {
            hit->pos = (int64_t)p->pos; hit->sa = p->sa; hit->type = p->type; hit->strand = p->strand; hit->mapq = p->mapq;
            hit->n_mm = p->n_mm; hit->n_gapo = p->n_gapo; hit->n_gape = p->n_gape; hit->ref_shift = p->ref_shift; hit->score = p->score;
            hit->c1 = p->c1; hit->c2 = p->c2; hit->nm = p->nm; hit->n_cigar = p->n_cigar; hit->n_multi = p->n_multi;
            hit->flag = flag; hit->seqid = seqid; hit->nn = nn; memcpy(hit->cigar, p->cigar, sizeof hit->cigar);
        }
    } else {
        int i;
        sputs(out, p->name); sputs(out, "\t4\t*\t0\t0\t*\t*\t0\t0\t");
        if (!p->strand) for (i = 0; i < p->len; ++i) sputc(out, "ACGTN"[p->seq[i]]);
        else for (i = 0; i < p->len; ++i) sputc(out, "ACGTN"[p->rseq[i]]);
        sputc(out, '\t');
        if (p->qual) { if (!p->strand) sputs(out, p->qual); else for (i = p->len - 1; i >= 0; --i) sputc(out, p->qual[i]); }
        else sputc(out, '*');
        sputc(out, '\n');
        if (hit) { hit->flag = 4; hit->c1 = p->c1; hit->c2 = p->c2; hit->strand = p->strand; }
    }
}

/* FASTQ/FASTA reader: name up to first whitespace, trailing /1 /2 stripped
 * (upstream bwa_read_seq, bwaseqio.c) */
static read_t *load_reads(const char *path, int64_t *n_out)
{
    size_t n, i = 0; int64_t cnt = 0, cap = 0;
    char *buf = read_file(path, &n);
    read_t *R = NULL;
    if (!buf) { FAIL("cannot read %s", path); return NULL; }
    while (i < n) {
        read_t *r; size_t s, e; int fq, t;
        while (i < n && buf[i] != '@' && buf[i] != '>') ++i;
        if (i >= n) break;
        fq = buf[i] == '@'; ++i; s = i;
        while (i < n && !isspace((unsigned char)buf[i])) ++i;
        if (cnt == cap) { cap = cap ? cap * 2 : 1024; R = (read_t *)realloc(R, sizeof(read_t) * (size_t)cap); }
        r = &R[cnt++]; memset(r, 0, sizeof *r);
        r->name = strndup(buf + s, i - s);
        t = (int)strlen(r->name);
        if (t > 2 && r->name[t - 2] == '/' && (r->name[t - 1] == '1' || r->name[t - 1] == '2')) r->name[t - 2] = 0;
        while (i < n && buf[i] != '\n') ++i;
        ++i; s = i;
        /* sequence: lines until '+' (fastq) or '>' / EOF (fasta) */
        { str_t sq = {0, 0, 0};
          while (i < n && buf[i] != (fq ? '+' : '>')) { if (isgraph((unsigned char)buf[i])) sputc(&sq, buf[i]); ++i; }
          r->len = (int)sq.l;
          r->seq = (uint8_t *)malloc(sq.l + 1); r->rseq = (uint8_t *)malloc(sq.l + 1);
          for (e = 0; e < sq.l; ++e) r->seq[e] = nt4(sq.s[e]);
          for (e = 0; e < sq.l; ++e) { uint8_t c = r->seq[sq.l - 1 - e]; r->rseq[e] = c > 3 ? c : 3 - c; }
          free(sq.s); }
        if (fq && i < n) {
            while (i < n && buf[i] != '\n') ++i;
            ++i;
            { str_t ql = {0, 0, 0};
              while (i < n && (int)ql.l < r->len) { if (isgraph((unsigned char)buf[i])) sputc(&ql, buf[i]); ++i; }
              r->qual = ql.s ? ql.s : strdup(""); }
        }
    }
    free(buf);
    if (!R) R = (read_t *)calloc(1, sizeof(read_t));   /* an input without reads is not an error: upstream prints the header and stops */
    *n_out = cnt;
    return R;
}

/* multi-process runs shard the reads but must keep ONE tie-break stream in input order: a shard
 * starts the stream after the draws its predecessors consumed and reports where it stopped. */
static uint64_t g_draws_before = 0, g_draws_after = 0;
void orc_set_rng_offset(uint64_t draws_before) { g_draws_before = draws_before; }
uint64_t orc_get_rng_draws(void) { return g_draws_after; }

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
/* stage times of the last orc_map_fastq: FASTQ parse, aln stage, samse stage up to the per-read records, SAM text + file */
static double g_times[4] = {0, 0, 0, 0};
void orc_last_times(double out[4]) { int j; for (j = 0; j < 4; ++j) out[j] = g_times[j]; }

int64_t orc_map_fastq(const orc_index_t *ix, const orc_opt_t *o, const char *fastq, const char *sam_out,
                      const char *sai_out, orc_hit_t *hits, int64_t hits_cap, int n_threads,
                      double *t_aln_s, double *t_samse_s)
{
    int64_t n, i;
    const double tp = now_s();
    read_t *R = load_reads(fastq, &n);
    FILE *fo, *fs = NULL;
    orc_rng_t rng;
    double t0, t1, t2, tm;
    str_t *lines;
    if (!R) return -1;
    if (n_threads < 1) n_threads = 1;
    t0 = now_s();
    /* ---- aln stage (upstream bwa_aln_core): independent per read ---- */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads)
#endif
    for (i = 0; i < n; ++i) {
        read_t *r = &R[i];
        int cap = 64, na;
        r->aln = (orc_aln_t *)malloc(sizeof(orc_aln_t) * (size_t)cap);
        na = orc_aln_one(ix, o, r->len, r->seq, r->aln, cap, NULL, NULL);
        if (na > cap) { /* rerun with exact capacity so the gapped de-dup sees every stored hit */
            cap = na; r->aln = (orc_aln_t *)realloc(r->aln, sizeof(orc_aln_t) * (size_t)cap);
            na = orc_aln_one(ix, o, r->len, r->seq, r->aln, cap, NULL, NULL);
        }
        r->n_aln = na;
    }
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    stats_flush();
    t1 = now_s();
    if (sai_out) {
        fs = fopen(sai_out, "wb");
        if (!fs) { FAIL("cannot write %s", sai_out); return -1; }
        for (i = 0; i < n; ++i) { int32_t na = R[i].n_aln; fwrite(&na, 4, 1, fs); fwrite(R[i].aln, sizeof(orc_aln_t), (size_t)na, fs); }
        fclose(fs);
    }
    /* ---- samse stage (upstream bwa_sai2sam_se_core) ---- */
    orc_srand48(&rng, 11);
    { uint64_t d, x0; for (d = 0; d < g_draws_before; ++d) rng_step(&rng);
      g_draws_after = g_draws_before; x0 = rng.x;
      for (i = 0; i < n; ++i) { /* one sequential RNG stream over reads */
          orc_rng_t before = rng; uint64_t used = 0;
          aln2seq(&R[i], o->n_occ, &rng);
          while (before.x != rng.x) { rng_step(&before); ++used; }
          g_draws_after += used;
      }
      (void)x0; }
    lines = (str_t *)calloc((size_t)n + 1, sizeof(str_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads)
#endif
    for (i = 0; i < n; ++i) {
        read_t *s = &R[i];
        int j, k, strand = 0, units = s->score; /* profile mode: units == score */
        if (s->type != 0) {
            s->pos = sa2pos(ix, s->sa, s->len + s->ref_shift, &strand);   /* upstream bwa_cal_pac_pos_core */
            s->strand = strand;
            s->mapq = approx_mapq(s, o, units);
            if (s->pos == (uint64_t)-1) s->type = 0;
        }
        for (j = k = 0; j < s->n_multi; ++j) {                              /* upstream bwa_cal_pac_pos */
            multi_t *q = s->multi + j;
            q->pos = sa2pos(ix, q->pos, s->len + q->ref_shift, &strand);
            q->strand = strand;
            if (q->pos != s->pos && q->pos != (uint64_t)-1) s->multi[k++] = *q;
        }
        s->n_multi = k;
        for (j = k = 0; j < s->n_multi; ++j) {                              /* upstream bwa_refine_gapped */
            multi_t *q = s->multi + j;
            if (q->gap) {
                q->n_cigar = refine_gapped(ix, s->len, q->strand ? s->rseq : s->seq, q->ref_shift, &q->pos, q->cigar);
                if (q->n_cigar) s->multi[k++] = *q;
            } else s->multi[k++] = *q;
        }
        s->n_multi = k;
        if (s->type != 0 && s->n_gapo) {
            s->n_cigar = refine_gapped(ix, s->len, s->strand ? s->rseq : s->seq, s->ref_shift, &s->pos, s->cigar);
            if (s->n_cigar == 0) s->type = 0;
        }
        if (s->type != 0) s->md = cal_md(ix, s->n_cigar, s->cigar, s->len, s->pos, s->strand ? s->rseq : s->seq, &s->nm);
    }
    tm = now_s();          /* per-read records complete; what follows is SAM text (timed apart for the bench's two scopes) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads)
#endif
    for (i = 0; i < n; ++i) print_sam(ix, o, &R[i], &lines[i], (hits && i < hits_cap) ? &hits[i] : NULL);
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    stats_flush();
    fo = fopen(sam_out, "wb");
    if (!fo) { FAIL("cannot write %s", sam_out); return -1; }
    for (i = 0; i < ix->n_seqs; ++i) fprintf(fo, "@SQ\tSN:%s\tLN:%d\n", ix->anns[i].name, ix->anns[i].len);
    for (i = 0; i < n; ++i) { fwrite(lines[i].s, 1, lines[i].l, fo); free(lines[i].s); }
    fclose(fo);
    t2 = now_s();
    if (t_aln_s) *t_aln_s = t1 - t0;
    if (t_samse_s) *t_samse_s = t2 - t1;
    g_times[0] = t0 - tp; g_times[1] = t1 - t0; g_times[2] = tm - t1; g_times[3] = t2 - tm;
    for (i = 0; i < n; ++i) { read_t *r = &R[i]; free(r->name); free(r->seq); free(r->rseq); free(r->qual); free(r->aln); free(r->md); free(r->multi); }
    free(R); free(lines);
    return n;
}
