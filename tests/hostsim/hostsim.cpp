// hostsim.cpp -- TEST HARNESS ONLY (never linked into libparasuite_hip.so).
// Runs the per-lane device logic of para-suite_amd/csrc/ps_core.h on the host,
// lane by lane, so that the CPU test tier can compare the exact kernel state
// machine with the oracle without a GPU.  The index blocks are packed here
// from a BWT symbol string supplied by the test (the oracle's), which also
// pins the block layout.
#include "../../para-suite_amd/csrc/ps_core.h"
#include "../../para-suite_amd/csrc/ps_narrow.h"
#include "../../para-suite_amd/csrc/ps_model.h"
#include <vector>
#include <cstring>
#include <cstdlib>
using namespace ps;

struct SimIndex {
    std::vector<OccBlock> blocks; std::vector<uint32_t> sa; std::vector<uint8_t> pac, jump_raw;
    IndexView v;
};

extern "C" {

void *hs_index_new(const uint8_t *bwt_syms, uint64_t n, uint64_t primary, const uint64_t *sa_samples, uint64_t n_sa,
                   int sa_intv, const uint8_t *pac, uint64_t l_pac)
{
    SimIndex *s = new SimIndex();
    uint32_t nb = (uint32_t)(n / PS_BLK_SYMS + 1), cnt[4] = {0, 0, 0, 0};
    s->blocks.resize(nb);
    for (uint32_t b = 0; b < nb; ++b) {
        uint64_t beg = (uint64_t)b * PS_BLK_SYMS, m = beg < n ? (n - beg < PS_BLK_SYMS ? n - beg : PS_BLK_SYMS) : 0;
        blk_pack(s->blocks[b], bwt_syms + (beg < n ? beg : 0), (int)m, cnt);
        for (uint64_t t = 0; t < m; ++t) ++cnt[bwt_syms[beg + t]];
    }
    s->sa.assign(n_sa + (n_sa + 31) / 32, 0);               // low words, then the bit-32 plane (the product's layout)
    for (uint64_t i = 0; i < n_sa; ++i) {
        s->sa[i] = (uint32_t)sa_samples[i];
        if (i && ((sa_samples[i] >> 32) & 1)) s->sa[n_sa + (i >> 5)] |= 1u << (i & 31);
    }
    s->pac.assign(pac, pac + l_pac / 4 + 1);
    IndexView &v = s->v;
    v.blocks = s->blocks.data(); v.sa = s->sa.data(); v.sa_hi = s->sa.data() + n_sa; v.pac = s->pac.data();
    v.seq_len = (bwtint)n; v.primary = (bwtint)primary; v.l_pac = (bwtint)l_pac;
    v.L2[0] = 0; for (int c = 0; c < 4; ++c) v.L2[c + 1] = v.L2[c] + cnt[c];
    v.n_blocks = nb; v.n_sa = (uint32_t)n_sa; v.sa_intv = sa_intv;
    return s;
}
void hs_index_free(void *p) { delete (SimIndex *)p; }

// the jump table of the index (ps_core.h), filled by the product's own slot function, level after level; levels < 0: as many
// as the product would take for this text; 0: none.  Returns the number of levels.
int hs_index_jump(void *p, int levels)
{
    SimIndex *s = (SimIndex *)p;
    if (levels < 0) levels = jump_levels_for(s->v.seq_len);
    if (levels > 12) levels = 12;                          // host memory
    s->v.jump = nullptr; s->v.jump_levels = 0;
    if (levels == 0) return 0;
    BtArgs a; memset(&a, 0, sizeof a); a.ix = s->v;
    BtHot h; (void)bt_hot_make(a, h);
    s->jump_raw.assign(jump_words(levels) * 4 + 16, 0);
    uint32_t *t = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(s->jump_raw.data()) + 15) & ~(uintptr_t)15);
    for (int d = 0; d < levels; ++d)
        for (uint32_t x = 0; x < (1u << (2 * d)); ++x) jump_fill_slot(h, t, s->v.seq_len, d, x);
    s->v.jump = t; s->v.jump_levels = levels;
    return levels;
}
// a slot of it: the eight words of string `sidx` of level d
void hs_jump_slot(void *p, int d, uint32_t sidx, uint32_t out[8])
{
    SimIndex *s = (SimIndex *)p;
    memcpy(out, s->v.jump + (size_t)(jump_level_off(d) + sidx) * PS_JUMP_SLOT_WORDS, 32);
}

uint32_t hs_occ(void *p, int64_t k, int c) // Occ(k,c) through the block code, k in [-1, n]
{
    SimIndex *s = (SimIndex *)p; LaneStats st; memset(&st, 0, sizeof st);
    uint32_t ok, ol;
    // occ_pair1 returns Occ(k'-1) and Occ(l): ask for k' = k+1
    occ_pair1(s->v, (bwtint)(k + 1), (bwtint)(k < 0 ? 0 : k), c, ok, ol, st);
    return ok;
}
uint64_t hs_sa(void *p, uint64_t row)
{
    SimIndex *s = (SimIndex *)p; LaneStats st; memset(&st, 0, sizeof st);
    bwtint r = (bwtint)row; uint32_t steps = 0;
    while (sa_walk_step(s->v, r, steps, st)) {}
    return (bwtint)steps + sa_sample(s->v, r / (bwtint)s->v.sa_intv);
}

// the estimated best score of every read for the NEXT hs_aln call (ps_narrow.h, nt_tail: children that can only matter if the best
// hit is worse than it are not stored; a read whose estimate fails starts over); nullptr: none.  The tests hand in true, too low,
// too high and random values: the hits must not depend on them.
static const uint8_t *g_est = nullptr;
void hs_set_est(const uint8_t *est) { g_est = est; }

// Width stage + backtracking stage for n_reads reads of one length, simulated with n_lanes lanes.
// codes: [n_reads][len] (0..3, 4 = N).  Outputs: w_out [len+1][n_reads] (pre-shadow), cwb, alns.
int hs_aln(void *p, const Model *md, int n_reads, int len, const uint8_t *codes, int n_lanes, int pool_cap, int aln_cap, int force_wide, int n_big,
           uint32_t *w_out, uint8_t *cwb_out, uint8_t *cswb_out, AlnRec *alns, int32_t *n_aln, uint8_t *status, KStats *ks)
{
    SimIndex *s = (SimIndex *)p;
    int n_bw = (len + 15) / 16, n_mw = (len + 31) / 32;
    std::vector<uint32_t> bases((size_t)n_bw * n_reads, 0), nmask((size_t)n_mw * n_reads, 0);
    for (int r = 0; r < n_reads; ++r)
        for (int j = 0; j < len; ++j) {
            int c = codes[(size_t)r * len + j];
            if (c > 3) nmask[(size_t)(j >> 5) * n_reads + r] |= 1u << (j & 31);
            else bases[(size_t)(j >> 4) * n_reads + r] |= (uint32_t)c << (2 * (j & 15));
        }
    int seed_len = md->seed_len;
    std::vector<uint32_t> w((size_t)(len + 1) * n_reads);
    std::vector<uint32_t> cwb((size_t)lm_ncw(len) * n_reads, 0), cswb((size_t)(lm_ncsw(seed_len) + 1) * n_reads, 0);
    auto put = [&](std::vector<uint32_t> &v, int pos, int r, uint8_t byte) { v[(size_t)(pos >> 2) * n_reads + r] |= (uint32_t)byte << (8 * (pos & 3)); };
    LaneStats st; memset(&st, 0, sizeof st);
    for (int r = 0; r < n_reads; ++r) {                 // width kernel body
        WChain A, B; wchain_init(s->v, A); wchain_init(s->v, B);
        for (int i = 0; i < len; ++i) {
            uint32_t wv; uint8_t cb;
            wchain_step(s->v, A, read_base(bases.data(), nmask.data(), n_reads, r, len - 1 - i), wv, cb, i == 0, st);
            w[(size_t)i * n_reads + r] = wv; put(cwb, i, r, cb);
            if (md->use_seed && i < seed_len) {
                wchain_step(s->v, B, read_base(bases.data(), nmask.data(), n_reads, r, seed_len - 1 - i), wv, cb, i == 0, st);
                put(cswb, i, r, cb);
            }
        }
        w[(size_t)len * n_reads + r] = 0; put(cwb, len, r, cw_pack(A.bid + 1, false));
        if (md->use_seed) put(cswb, seed_len, r, cw_pack(B.bid + 1, false));
    }
    if (w_out) memcpy(w_out, w.data(), w.size() * 4);
    (void)cwb_out; (void)cswb_out;
    BtArgs a; memset(&a, 0, sizeof a);
    a.ix = s->v; a.md = *md; a.n_reads = n_reads; a.len = len; a.n_lanes = n_lanes;
    a.bases = bases.data(); a.nmask = nmask.data(); a.n_bw = n_bw; a.n_mw = n_mw;
    a.w = w.data(); a.cwb = cwb.data(); a.cswb = cswb.data();
    a.alns = alns; a.aln_cap = aln_cap; a.n_aln = n_aln; a.status = status;
    const bool wide = pool_cap > 65535 || force_wide;
    a.est = g_est; a.cap_est = (g_est && md->profile && !wide) ? 1 : 0; g_est = nullptr;
    std::vector<uint8_t> pool((size_t)n_lanes * pool_cap * (wide ? sizeof(Entry) : sizeof(Entry16)));
    std::vector<uint32_t> heads((size_t)n_lanes * PS_MAX_BUCKETS);
    a.pool = pool.data(); a.pool_cap = (uint32_t)pool_cap; a.heads = heads.data(); a.wide = wide;
    const uint32_t big_cap = 65535;
    std::vector<std::vector<uint8_t>> big_slots;            // large stacks handed out on M_GROW (the kernel uses one arena + an atomic)
    a.n_big = wide ? 0 : (uint32_t)n_big; a.big_cap = big_cap;
    int lmb = lm_bytes(len, seed_len, md->n_buckets, wide);
    std::vector<uint8_t *> cur_pool(n_lanes, nullptr);
    std::vector<uint8_t> lm((size_t)n_lanes * lmb);
    std::vector<BtLane> lanes(n_lanes); std::vector<NLane> nlanes(n_lanes); std::vector<LaneStats> nst(n_lanes); std::vector<int> next(n_lanes);
    for (int t = 0; t < n_lanes; ++t) { memset(&lanes[t], 0, sizeof(BtLane)); lanes[t].mode = M_FETCH; nl_init(nlanes[t]); ls_init(nst[t]); next[t] = t; }
    const bool nb32 = md->n_buckets <= 32 && (n_lanes & 1) == 0;        // as the product chooses; odd lane counts: the general 64-bit form on any model
    BtHot h;
    a.big_cap = wide ? 0 : big_cap;
    if (!bt_hot_make(a, h)) return -2;
    bool any = true;
    while (any) {                                        // lock-step over lanes, like a wave
        any = false;
        for (int t = 0; t < n_lanes; ++t) {
            BtMem m; uint8_t *mine = lm.data() + (size_t)t * lmb;
            bt_mem_bind(m, mine, len, seed_len);
            uint8_t *priv = pool.data() + (size_t)t * pool_cap * (wide ? sizeof(Entry) : sizeof(Entry16));
            if (!wide) {                                     // the narrow tiers: packed lane state (ps_narrow.h)
                NLane &L = nlanes[t];
                int mode = nl_mode(L.ctl);
                if (mode == M_EXIT) { m.pool = priv; m.heads = nullptr; if (nb32) nt_iter<true, true>(a, h, L, nst[t], m, -1, true); else nt_iter<true, false>(a, h, L, nst[t], m, -1, true); continue; }   // the kernel calls retired lanes too: must be a no-op
                if (mode == M_FETCH) { cur_pool[t] = priv; L.ctl &= ~NL_BIG; }      // read done: a large slot goes back, the next read starts on the private slice
                if (!cur_pool[t]) cur_pool[t] = priv;
                if (mode == M_GROW) {                        // what the kernel does wave-cooperatively
                    if (big_slots.size() < a.n_big) {
                        big_slots.emplace_back((size_t)big_cap * sizeof(Entry16));
                        memcpy(big_slots.back().data(), cur_pool[t], (size_t)nl_bump(L) * sizeof(Entry16));
                        cur_pool[t] = big_slots.back().data(); L.ctl = nl_set_mode(L.ctl | NL_BIG, M_EXPAND);
                    } else L.ctl = nl_set_mode(nl_set_status(L.ctl, RS_OVERFLOW_POOL), M_POP);
                    mode = nl_mode(L.ctl);
                }
                m.pool = cur_pool[t]; m.heads = nullptr;
                int fr = -1;                               // static hand-out here; the kernel deals reads from a queue
                if (mode == M_FETCH) { fr = next[t] < n_reads ? next[t] : n_reads; next[t] += n_lanes; }
                if (nb32) nt_iter<true, true>(a, h, L, nst[t], m, fr, (t & 1) != 0 || (nst[t].iters & 3) == 0);
                else nt_iter<true, false>(a, h, L, nst[t], m, fr, (t & 1) != 0 || (nst[t].iters & 3) == 0);
                any = true;
                continue;
            }
            BtLane &L = lanes[t];
            if (L.mode == M_EXIT) { BtMem mm{}; bt_iter(a, h, L, mm, -1, true); continue; }
            m.pool = priv;
            m.heads = a.heads + (size_t)t * PS_MAX_BUCKETS;
            int fr = -1;
            if (L.mode == M_FETCH) { fr = next[t] < n_reads ? next[t] : n_reads; next[t] += n_lanes; }
            bt_iter(a, h, L, m, fr, (t & 1) != 0 || L.mode != M_HIT || (L.st.iters & 3) == 0);
            any = true;
        }
    }
    if (!wide) for (int t = 0; t < n_lanes; ++t) lanes[t].st = nst[t];
    if (ks) {
        memset(ks, 0, sizeof *ks);
        for (int t = 0; t < n_lanes; ++t) {
            ks->occ_pairs += lanes[t].st.pairs; ks->occ_same_blk += lanes[t].st.same; ks->nodes += lanes[t].st.nodes;
            ks->pushes += lanes[t].st.pushes; ks->pops += lanes[t].st.pops; ks->iters += lanes[t].st.iters;
            ks->exact_steps += lanes[t].st.exact;
        }
        ks->occ_pairs += st.pairs; ks->occ_same_blk += st.same;
    }
    return 0;
}

// banded global alignment of query codes against pac[rb, rb+tlen)
int hs_banded(void *p, int qlen, const uint8_t *q, uint64_t rb, int tlen, int w, uint32_t *cigar, int cap)
{
    SimIndex *s = (SimIndex *)p;
    std::vector<int32_t> H(qlen + 2), E(qlen + 2);
    int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    std::vector<uint8_t> z((size_t)n_col * tlen + 1);
    return banded_global(qlen, [&](int j) { return (int)q[j]; }, tlen, s->v.pac, rb, w, H.data(), E.data(), 1, z.data(), 1, cigar, cap);
}

int hs_model_stock(const char *n_arg, int len, Model *out)
{
    Options o; std::string err; set_stock_n(o, n_arg);
    return make_model(o, len, *out, err) ? 0 : -1;
}
int hs_model_profile(const double *P, double ins, double del, int x, int len, Model *out)
{
    Options o; std::string err; profile_costs(o, P, ins, del, x);
    return make_model(o, len, *out, err) ? 0 : -1;
}
// 33-bit rows: the packed representations must round-trip values above 2^32 (no genome that large fits a CPU test)
int hs_unit_rows33(void)
{
    const bwtint big[] = {0ull, 1ull, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, 0x1ABCDEF12ull, PS_MAX_ROWS};
    for (bwtint k : big) for (bwtint l : big) {
        if (l < k) continue;
        // narrow stack entry (ps_narrow.h): words round-trip through a push and a pop
        std::vector<uint8_t> lmem(4096, 0), pool(64 * sizeof(Entry16), 0);
        BtLane L; memset(&L, 0, sizeof L); L.free_head = PS_NIL; L.cap = 64;
        BtMem m; bt_mem_bind(m, lmem.data(), 50, 32); m.pool = pool.data(); m.heads = nullptr;
        BtArgs a0; memset(&a0, 0, sizeof a0); a0.md.profile = 1; a0.md.n_buckets = 64; a0.pool_cap = 64;
        a0.ix.L2[1] = 0x90000000ull; a0.ix.L2[2] = 0x120000000ull; a0.ix.L2[3] = 0x1B0000000ull; a0.ix.seq_len = 0x1F0000123ull;
        BtHot a; if (!bt_hot_make(a0, a)) return 7;
        {
            NLane N; nl_init(N); nt_heads_init(m, 64);
            const uint32_t wa = 17u | (17u << 8) | (3u << 16) | (((uint32_t)ST_D | (2u << 2) | (5u << 5)) << 24), wb = 7u | (6u << 3) | (3u << 6) | (9u << 9);
            nt_push(N, m, (uint32_t)k, (uint32_t)l, wa, wb);
            N.kr = N.lr = N.wa = N.wb = 0;
            nt_pop<false>(N, m);
            if (N.kr != (uint32_t)k || N.lr != (uint32_t)l || N.wa != wa || N.wb != wb || nl_n_stack(N) != 0 || N.bm0 != 0 || N.bm1 != 0) return 1;
            if (nw_i(N.wa) != 17 || nw_ldp(N.wa) != 17 || nw_mm(N.wa) != 3 || nw_state(N.wa) != ST_D || nw_gapo(N.wa) != 2 || nw_gape(N.wa) != 5 ||
                nw_ins(N.wb) != 7 || nw_del(N.wb) != 6 || nw_c(N.wb) != 3 || nw_score(N.wb) != 9) return 2;
            for (uint32_t c = 0; c < 4; ++c) if (nt_base(a, c) != a0.ix.L2[c]) return 8;
            if (nt_base(a, NW_ROOT_C) != 0x100000000ull) return 9;      // the root: base + low word of n = n
        }
        // the wide entry
        const BtHot &aw = a;
        std::vector<uint8_t> wpool(64 * sizeof(Entry), 0); std::vector<uint32_t> heads(PS_MAX_BUCKETS, 0);
        m.pool = wpool.data(); m.heads = heads.data();
        memset(&L, 0, sizeof L); L.free_head = PS_NIL; L.max_units = 100;
        bt_push_wide(aw, L, m, true, 21, k, l, 4, 1, 2, 3, 0, ST_I, true, 11, 11);
        bt_pop(aw, L, m);
        if (L.k != k || L.l != l || L.i != 21) return 3;
    }
    for (bwtint sidx : big) {                                  // block addressing
        int off = -1; const uint32_t b = blk_of(sidx, off);
        if ((bwtint)b != sidx / PS_BLK_SYMS || (bwtint)off != sidx % PS_BLK_SYMS) return 4;
    }
    if (width32(0, 0x17FFFFFFFull) != 0xFFFFFFFFu || width32(5, 5) != 1u || width32(0x100000000ull, 0x100000009ull) != 10u) return 5;
    // sampled SA with the bit-32 plane
    uint32_t sa[4 + 1] = {0xFFFFFFFFu, 7u, 0x80000000u, 5u, 0u}; sa[4] = (1u << 3) | (1u << 1);
    IndexView v; memset(&v, 0, sizeof v); v.sa = sa; v.sa_hi = sa + 4;
    if (sa_sample(v, 0) != ~(bwtint)0 || sa_sample(v, 1) != 0x100000007ull || sa_sample(v, 2) != 0x80000000ull || sa_sample(v, 3) != 0x100000005ull) return 6;
    return 0;
}
size_t hs_sizeof_model(void) { return sizeof(Model); }
size_t hs_sizeof_alnrec(void) { return sizeof(AlnRec); }
}
