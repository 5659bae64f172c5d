// ps_model.h -- host-side option handling and cost model (header only).
//
// Options mirror the command lines the reference emits:
//   bwa aln -t T -n <mm> <ref> <fq> -f <sai>            BWAMapping.java:51-61
//   bwa parasuite -t T -X <mm> -p EP -g IP <ref> <fq>   PARAsuiteMapping.java:63-77
//   bwa samse <ref> <sai> <fq> -f <sam>                 PARAsuiteMapping.java:85-92
// Defaults are upstream BWA 0.7.x `aln`/`samse` defaults.  The PAR-CLIP
// substitution-aware penalty is this project's own rule (the fork's source is
// not in the reference tree); it is stated in
// profile_costs() below and restated independently in oracle/ps_oracle.c.
#pragma once
#include "ps_types.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace ps {

struct Options {
    int max_diff = -1;      // aln -n INT
    double fnr = 0.04;      // aln -n FLOAT
    int max_gapo = 1, max_gape = 6, mode_gape = 1;
    int indel_end_skip = 5, max_del_occ = 10, max_entries = 2000000;
    int seed_len = 32, max_seed_diff = 2, max_top2 = 30;
    int s_mm = 3, s_gapo = 11, s_gape = 4;
    int n_occ = 3;          // samse -n
    // PAR-CLIP profile mode
    int profile = 0, unit = 1, x_avg_mm = -1;
    int sub_cost[16] = {0}; // [ref*4+read], read orientation
    int n_cost = 1, gapo_ins_cost = 1, gapo_del_cost = 1, gape_cost = 1;
};

// smallest k whose Poisson(l*err) upper tail drops below thres (BWA's per-length difference budget)
inline int cal_maxdiff(int l, double err, double thres)
{
    double elambda = std::exp(-l * err), sum = elambda, y = 1.0;
    unsigned x = 1;
    for (int k = 1; k < 1000; ++k) {
        y *= l * err;
        x *= (unsigned)k;
        sum += elambda * y / (int)x;
        if (1.0 - sum < thres) return k;
    }
    return 2;
}

inline int budget_diffs(const Options &o, int len)
{
    if (o.profile) return o.x_avg_mm >= 0 ? o.x_avg_mm : cal_maxdiff(len, 0.02, 0.04);
    return o.fnr > 0.0 ? cal_maxdiff(len, 0.02, o.fnr) : o.max_diff;
}

// `-n` argument of `bwa aln`: a value containing '.' is a false-negative rate, otherwise a count
inline void set_stock_n(Options &o, const char *s)
{
    if (std::strchr(s, '.')) { o.fnr = std::atof(s); o.max_diff = -1; }
    else { o.max_diff = std::atoi(s); o.fnr = -1.0; }
}

// Our PAR-CLIP cost rule.  P[a*4+b] = P(read b | ref a) in read orientation
// (ErrorProfiling.java:504-531).  One average mismatch = U = 8 units, where
// "average" is the mean log-probability of the 12 substitution types.
inline void profile_costs(Options &o, const double P[16], double ins_rate, double del_rate, int x_avg_mm)
{
    const int U = 8;
    double L[16], lbar = 0.0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) {
        double p = P[a * 4 + b];
        if (!(p > 1e-9)) p = 1e-9;
        if (p > 1.0) p = 1.0;
        L[a * 4 + b] = std::log(p);
        if (a != b) lbar += L[a * 4 + b];
    }
    lbar /= 12.0;
    if (lbar > -1e-6) lbar = -1e-6;
    o.profile = 1; o.unit = U; o.x_avg_mm = x_avg_mm;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) {
        int c = 0;
        if (a != b) {
            c = (int)std::floor(U * L[a * 4 + b] / lbar + 0.5);
            if (c < 1) c = 1;
            if (c > 4 * U) c = 4 * U;
        }
        o.sub_cost[a * 4 + b] = c;
    }
    o.n_cost = U;
    const double rate[2] = {ins_rate, del_rate};
    int *dst[2] = {&o.gapo_ins_cost, &o.gapo_del_cost};
    for (int t = 0; t < 2; ++t) {
        int c;
        if (rate[t] > 0.0 && rate[t] < 1.0) {
            c = (int)std::floor(U * std::log(rate[t]) / lbar + 0.5);
            if (c < U) c = U;
            if (c > 8 * U) c = 8 * U;
        } else c = (int)std::floor(U * 11.0 / 3.0 + 0.5);
        *dst[t] = c;
    }
    o.gape_cost = (int)std::floor(U * 4.0 / 3.0 + 0.5);
}

// .errorprofile: 16 whitespace-separated doubles, row = reference base A,C,G,T (ErrorProfiling.java:504-531);
// .indelprofile: "ins\tdel" on one line (ErrorProfiling.java:545-591).  The Java caller may pass a literal
// null for the indel profile (Main.java:193-203 leaves it unset) -> stock gap ratio.
inline bool read_profile_files(const char *ep, const char *ip, double P[16], double &ins, double &del, std::string &err)
{
    FILE *f = std::fopen(ep, "r");
    if (!f) { err = std::string("cannot open error profile ") + ep; return false; }
    for (int i = 0; i < 16; ++i) {
        char tok[64];
        if (std::fscanf(f, "%63s", tok) != 1) { std::fclose(f); err = std::string("error profile needs 16 values: ") + ep; return false; }
        P[i] = std::strtod(tok, nullptr);
    }
    std::fclose(f);
    ins = del = 0.0;
    if (ip && ip[0] && std::strcmp(ip, "null") != 0) {
        f = std::fopen(ip, "r");
        if (!f) { err = std::string("cannot open indel profile ") + ip; return false; }
        char a[64], b[64];
        if (std::fscanf(f, "%63s %63s", a, b) == 2) { ins = std::strtod(a, nullptr); del = std::strtod(b, nullptr); }
        std::fclose(f);
    }
    return true;
}

// search model for reads of one length
inline bool make_model(const Options &o, int len, Model &m, std::string &err)
{
    std::memset(&m, 0, sizeof m);
    m.len = len; m.profile = o.profile ? 1 : 0;
    m.max_gape = o.max_gape; m.mode_gape = o.mode_gape; m.indel_end_skip = o.indel_end_skip;
    m.max_del_occ = o.max_del_occ; m.max_entries = o.max_entries; m.max_seed_diff = o.max_seed_diff; m.max_top2 = o.max_top2;
    m.use_seed = len > o.seed_len; m.seed_len = m.use_seed ? o.seed_len : 0;
    int max_cost = 0;
    if (!o.profile) {
        int md = budget_diffs(o, len);
        m.max_gapo = o.max_gapo < md ? o.max_gapo : md;
        for (int s = 0; s < 5; ++s) for (int c = 0; c < 4; ++c) { int mm = s != c; m.u_mm[s][c] = (uint8_t)mm; m.s_mm[s][c] = (uint8_t)(mm * o.s_mm); }
        m.u_gapo_ins = m.u_gapo_del = 1; m.s_gapo_ins = m.s_gapo_del = o.s_gapo;
        m.u_gape = o.mode_gape ? 1 : 0; m.s_gape = o.s_gape;
        m.s_stop = o.s_mm; m.u_tight = 1; m.c_min = 1; m.max_units = md;
        // score buckets: only entries whose units fit the budget are ever pushed, so the largest score on a
        // stack is max{a*s_mm + b*s_gapo + c*s_gape : a+b+c(u_gape) <= md, b <= max_gapo, c <= max_gape, c only with b}
        int top = 0;
        for (int bb = 0; bb <= m.max_gapo; ++bb)
            for (int cc = 0; cc <= (bb ? o.max_gape : 0); ++cc) {
                const int used = bb + (o.mode_gape ? cc : 0);
                if (used > md) continue;
                const int sc = (md - used) * o.s_mm + bb * o.s_gapo + cc * o.s_gape;
                if (sc > top) top = sc;
            }
        m.n_buckets = top + 1;
    } else {
        const int U = o.unit;
        m.max_gapo = o.max_gapo;
        m.c_min = 1 << 30;
        for (int s = 0; s < 5; ++s) for (int c = 0; c < 4; ++c) {
            // the search matches the reverse-complemented read against T: ref base = 3-c, read base = 3-s
            int cost = s == 4 ? o.n_cost : (s == c ? 0 : o.sub_cost[(3 - c) * 4 + (3 - s)]);
            m.u_mm[s][c] = m.s_mm[s][c] = (uint8_t)cost;
            if (cost > 0 && cost < m.c_min) m.c_min = cost;
            if (cost > max_cost) max_cost = cost;
        }
        m.u_gapo_ins = m.s_gapo_ins = o.gapo_ins_cost;
        m.u_gapo_del = m.s_gapo_del = o.gapo_del_cost;
        m.u_gape = m.s_gape = o.gape_cost;
        const int g[3] = {o.gapo_ins_cost, o.gapo_del_cost, o.gape_cost};
        for (int t = 0; t < 3; ++t) { if (g[t] < m.c_min) m.c_min = g[t]; if (g[t] > max_cost) max_cost = g[t]; }
        if (m.c_min < 1) m.c_min = 1;
        m.s_stop = U; m.u_tight = U;
        m.max_units = budget_diffs(o, len) * U;
        m.n_buckets = m.max_units + 1;      // profile mode: score == units, and unaffordable children are not pushed
        (void)max_cost;
    }
    m.inv_c_min = (65536 + m.c_min - 1) / m.c_min;
    for (int s = 0; s < 5; ++s) {
        m.u_mm_pk[s] = m.s_mm_pk[s] = 0;
        for (int c = 0; c < 4; ++c) { m.u_mm_pk[s] |= (uint32_t)m.u_mm[s][c] << (8 * c); m.s_mm_pk[s] |= (uint32_t)m.s_mm[s][c] << (8 * c); }
    }
    if (m.n_buckets > PS_MAX_BUCKETS) { err = "score range exceeds PS_MAX_BUCKETS; lower -n/-X"; return false; }
    if (len > PS_MAX_LEN) { err = "read longer than PS_MAX_LEN"; return false; }
    if (m.max_gapo > 7 || m.max_gape > 7) { err = "gap limits above 7 are not supported by the packed stack entry"; return false; }
    if (m.max_units / m.c_min > 126) { err = "difference budget too large"; return false; }
    return true;
}

}  // namespace ps
