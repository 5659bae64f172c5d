// ps_narrow.h -- the hot loop of the narrow search tiers (16-byte stack entries): what `bwa aln` /
// `bwa parasuite` compute for /root/reference/src/src/mapping/PARAsuiteMapping.java:63-77 and
// BWAMapping.java:51-61, one read per lane.  Same search, same visiting order and same results as the wide
// tier's bt_iter in ps_core.h (and as oracle/ps_oracle.c); what differs is how a lane holds its state:
//
//  * the current entry stays PACKED in four registers exactly as it sits on the stack (kr, lr, wa, wb), a
//    child's words are the parent's plus a constant, and a pop is a 16-byte load with nothing to unpack;
//  * SA intervals are stored relative to the rows of their first symbol c: rows [L2[c]+kr, L2[c]+lr] with
//    kr = Occ(k-1,c)+1 and lr = Occ(l,c) of the parent -- 32-bit words (one symbol has < 2^32 occurrences),
//    no 64-bit arithmetic and no L2 lookup when children are stored, one base add when the entry is used.
//    The root [0, n] is c = 4 (base = bit 32 of n, kr = 0, lr = low word of n);
//  * both Occ blocks of a step are always loaded (the block of row k-1 again when it is the block of row l:
//    an L1 hit) instead of one conditional load plus sixteen selects, and the root counts Occ(-1,.) = 0 from
//    block 0, whose running counts are zero;
//  * the lane state is packed into 14 registers (counts share words), statistics are a template parameter: the
//    production kernel carries no counters.
#pragma once
#include "ps_core.h"

namespace ps {

// ---- packed entry words ------------------------------------------------------------------------------------
// wa = i | last_diff_pos<<8 | n_mm<<16 | (state | n_gapo<<2 | n_gape<<5)<<24
// wb = n_ins | n_del<<3 | c<<6 | score<<9           (16 bits; the stack entry keeps `next` in the upper half)
PS_HD int nw_i(uint32_t wa) { return (int)(wa & 0xffu); }
PS_HD int nw_ldp(uint32_t wa) { return (int)((wa >> 8) & 0xffu); }
PS_HD int nw_mm(uint32_t wa) { return (int)((wa >> 16) & 0xffu); }
PS_HD int nw_state(uint32_t wa) { return (int)((wa >> 24) & 3u); }
PS_HD int nw_gapo(uint32_t wa) { return (int)((wa >> 26) & 7u); }
PS_HD int nw_gape(uint32_t wa) { return (int)(wa >> 29); }
PS_HD int nw_ins(uint32_t wb) { return (int)(wb & 7u); }
PS_HD int nw_del(uint32_t wb) { return (int)((wb >> 3) & 7u); }
PS_HD uint32_t nw_c(uint32_t wb) { return (wb >> 6) & 7u; }
PS_HD int nw_score(uint32_t wb) { return (int)((wb >> 9) & 0x7fu); }
static const uint32_t NW_KEEP = 0xFCFF0000u;       // of wa: n_mm and the gap counts (position, last_diff_pos and state are set per child)
static const uint32_t NW_C_MASK = 7u << 6;
static const uint32_t NW_ROOT_C = 4u;

// lane control word: mode | have_cur<<3 | status<<4 (3 bits) | on_big<<7 | n_aln<<16
PS_HD int nl_mode(uint32_t ctl) { return (int)(ctl & 7u); }
PS_HD uint32_t nl_set_mode(uint32_t ctl, int mode) { return (ctl & ~7u) | (uint32_t)mode; }
static const uint32_t NL_HAVE_CUR = 8u, NL_STATUS = 7u << 4, NL_BIG = 1u << 7;
PS_HD int nl_status(uint32_t ctl) { return (int)((ctl >> 4) & 7u); }
PS_HD uint32_t nl_set_status(uint32_t ctl, int s) { return (ctl & ~NL_STATUS) | ((uint32_t)s << 4); }
PS_HD int nl_n_aln(uint32_t ctl) { return (int)(ctl >> 16); }

struct NLane {
    uint32_t kr, lr, wa, wb;      // the current entry (in registers; "virtually" on the stack while NL_HAVE_CUR)
    uint32_t ctl;
    int r;
    uint32_t lim;                 // best_score (0xff: no hit yet) | max_units<<8 | len<<16 | min(best_cnt, 255)<<24
    uint32_t nsb;                 // n_stack | bump<<16 (live entries; first never-used slot)
    uint32_t n_phantom;           // children counted but not stored (low 24 bits) | the read's estimated best score << 24 (255: none; nt_tail)
    uint32_t fh;                  // free slot: the one popped last (0xffff: none)
    uint32_t bm0, bm1;            // non-empty score buckets 0..31 / 32..63 (narrow tiers have at most 64; two words, not one 64-bit value: with <= 32
                                  // buckets -- every default cost model -- the second word is never touched and costs no register moves)
    uint32_t rn0, rn1;            // N mask of a read of up to 64 bases (read orientation)
};
PS_HD int nl_best_score(const NLane &L) { return (int)(L.lim & 0xffu); }
PS_HD int nl_max_units(const NLane &L) { return (int)((L.lim >> 8) & 0xffu); }
PS_HD int nl_len(const NLane &L) { return (int)((L.lim >> 16) & 0xffu); }
PS_HD int nl_n_stack(const NLane &L) { return (int)(L.nsb & 0xffffu); }
PS_HD uint32_t nl_bump(const NLane &L) { return L.nsb >> 16; }
PS_HD uint32_t nl_phantom(const NLane &L) { return L.n_phantom & 0x00FFFFFFu; }
PS_HD int nl_est(const NLane &L) { return (int)(L.n_phantom >> 24); }

PS_HD void nl_init(NLane &L)
{
    L.kr = L.lr = L.wa = L.wb = 0; L.ctl = (uint32_t)M_FETCH; L.r = 0; L.lim = 0; L.nsb = 0;
    L.n_phantom = 0; L.fh = 0xffffu; L.bm0 = L.bm1 = 0; L.rn0 = L.rn1 = 0;
}
PS_HD void ls_init(LaneStats &st) { st.pairs = st.same = st.nodes = st.pushes = st.pops = st.iters = st.exact = st.lf = 0; }
// 24-bit multiply (full rate on the device; the general 32-bit one runs at a quarter of it): for the small budget arithmetic
PS_HD uint32_t ps_mul24(uint32_t x, uint32_t y)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __umul24(x, y);
#else
    return x * y;
#endif
}

// Diagnostic build only (make STAMPS=1): a wave's clock between the three parts of an iteration, taken where the
// wave has reconverged (nt_iter).  s_memtime drains the LDS counter too, so that build is slower than the product.
struct NtClock { unsigned long long tm[4]; unsigned long long t_last; };
#if defined(PS_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define PS_USTAMP(clk, slot) do { if (STATS && (clk)) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); (clk)->tm[slot] += t_ - (clk)->t_last; (clk)->t_last = t_; } } while (0)
#else
#define PS_USTAMP(clk, slot) do { } while (0)
#endif

// first row of symbol c's range (c = 4: the root's base)
PS_HD bwtint nt_base(const BtHot &h, uint32_t c)
{
    return (bwtint)sel4s(h.L2lo[0], h.L2lo[1], h.L2lo[2], h.L2lo[3], (int)c) | ((bwtint)((h.L2hi >> c) & 1u) << 32);
}
// budget units an entry has used: profile mode has units == score, stock counts every edit as one unit
PS_HD int nt_units(const BtHot &h, uint32_t wa, uint32_t wb)
{
    return h.profile() ? nw_score(wb) : nw_mm(wa) + nw_gapo(wa) + (h.mode_gape() ? nw_gape(wa) : 0);
}

// occurrences of every symbol among the first r (0..192) symbols of the block, plus the block's running counts:
// with A = |lo|, B = |hi|, C = |lo & hi| over the prefix the four counts are r-A-B+C, A-C, B-C, C
PS_HD void blk_count4b(const Blk &b, int r, uint32_t cnt[4])
{
    uint32_t A = 0, B = 0, C = 0;
    const int q = r >> 5;
    const uint32_t part = (1u << (r & 31)) - 1u;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const uint32_t m = j < q ? 0xFFFFFFFFu : (j == q ? part : 0u), lo = b.x[4 + j] & m, hi = b.x[10 + j] & m;
        A += ps_popc(lo); B += ps_popc(hi); C += ps_popc(lo & hi);
    }
    cnt[0] = b.x[0] + ((uint32_t)r - A - B + C);
    cnt[1] = b.x[1] + (A - C); cnt[2] = b.x[2] + (B - C); cnt[3] = b.x[3] + C;
}

#ifndef PS_OCC_COND
#define PS_OCC_COND 1       // 1: the block of row k-1 is loaded only where it is not the block of row l (exec-masked load +
#endif                     // sixteen selects); 0: always (measured 7 % slower: the redundant requests cost the L1 more than the selects cost)
// Occ(k-1, .) and Occ(l, .) of the interval (c, kr, lr): the memory operation of one search step
template <bool STATS>
PS_HD void nt_occ(const BtHot &h, uint32_t c, uint32_t kr, uint32_t lr, uint32_t ck[4], uint32_t cl[4], LaneStats &st)
{
    const bwtint base = nt_base(h, c);
    const bool need_k = c != NW_ROOT_C;
    int ol_ = 0, ok_ = 0;
    const uint32_t bl = blk_of(row_to_stored(h.primary, base + lr), ol_);
    uint32_t bk = blk_of(row_to_stored(h.primary, base + (bwtint)(need_k ? kr - 1u : 0u)), ok_);
    bk = need_k ? bk : 0u;                       // the root: Occ(-1,.) = 0 = the running counts of block 0
    const int rk = need_k ? ok_ + 1 : 0;
    Blk xl, xk;
    load_blk(h.blocks, bl, xl);
    if (PS_OCC_COND) {
        // both blocks are asked for before either is looked at, and the block of row l is counted first: the second one's
        // load (where there is one) has that long to arrive
        const bool other = bk != bl;
        if (other) load_blk(h.blocks, bk, xk);
        blk_count4b(xl, ol_ + 1, cl);
#pragma unroll
        for (int j = 0; j < 16; ++j) xk.x[j] = other ? xk.x[j] : xl.x[j];
    } else { load_blk(h.blocks, bk, xk); blk_count4b(xl, ol_ + 1, cl); }
    blk_count4b(xk, rk, ck);
    // Block reads of the steps that go through the Occ array (not the jump table's), counted in EVERY build: two adds, and the
    // timed kernel can say what it asked the memory for.  (They also decide the schedule the compiler picks for this function:
    // with them it waits for the blocks late and lets the table's loads run on, without them it waits right behind every load --
    // 1255 ms against 1334 ms per 10 M reads, profiles/r02_kernel_experiments.txt.  A measured property of hipcc 7.2, not a design.)
    ++st.pairs; if (need_k && bk == bl) ++st.same;
    (void)STATS;
}

// text symbols an entry has consumed (its depth in the tree of strings): read bases used, less the inserted ones, plus the deleted
PS_HD int nt_depth(int len, uint32_t wa, uint32_t wb) { return len - nw_i(wa) - (int)(wb & 7u) + (int)((wb >> 3) & 7u); }
// the same two results out of the jump table: the entry is at depth d < jump_levels and carries its string's index
PS_HD void nt_jump(const BtHot &h, int d, uint32_t sidx, uint32_t ck[4], uint32_t cl[4])
{
    const uint32_t *sl = h.jump + (size_t)(jump_level_off(d) + sidx) * PS_JUMP_SLOT_WORDS;
#ifdef __HIP_DEVICE_COMPILE__
    const ps_u32x4 a = *PS_AS_GLOBAL(ps_u32x4, sl), b = *PS_AS_GLOBAL(ps_u32x4, sl + 4);
    ck[0] = a.x; cl[0] = a.y; ck[1] = a.z; cl[1] = a.w; ck[2] = b.x; cl[2] = b.y; ck[3] = b.z; cl[3] = b.w;
#else
    for (int c = 0; c < 4; ++c) { ck[c] = sl[2 * c]; cl[c] = sl[2 * c + 1]; }
#endif
}
// one slot of the table (the builder: level d is filled after level d - 1).  The string's own interval is child (s & 3) of its
// parent's slot; the slot holds nt_occ of that interval, computed by nt_occ itself.
PS_HD void jump_fill_slot(const BtHot &h, uint32_t *table, bwtint seq_len, int d, uint32_t s)
{
    uint32_t c = NW_ROOT_C, kr = 0, lr = (uint32_t)seq_len;
    bool empty = false;
    if (d > 0) {
        const uint32_t *ps = table + (size_t)(jump_level_off(d - 1) + (s >> 2)) * PS_JUMP_SLOT_WORDS;
        c = s & 3u;
        empty = ps[2 * c] >= ps[2 * c + 1];
        kr = ps[2 * c] + 1u; lr = ps[2 * c + 1];
    }
    uint32_t ck[4] = {0, 0, 0, 0}, cl[4] = {0, 0, 0, 0};
    LaneStats st;
    if (!empty) nt_occ<false>(h, c, kr, lr, ck, cl, st);
    uint32_t *out = table + (size_t)(jump_level_off(d) + s) * PS_JUMP_SLOT_WORDS;
    for (int j = 0; j < 4; ++j) { out[2 * j] = ck[j]; out[2 * j + 1] = cl[j]; }
}

// base j of the reverse-complemented read (what the search consumes): 0..3, 4 = N
PS_HD int nt_seq_at(const BtMem &m, const NLane &L, int j, int len, int max_len)
{
    const int p = len - 1 - j;
    const uint32_t b = (m.rb[p >> 4] >> (2 * (p & 15))) & 3u;
    const uint32_t nw = lm_nmask_in_regs(max_len) ? (p < 32 ? L.rn0 : L.rn1) : m.rn[p >> 5];
    return ((nw >> (p & 31)) & 1u) ? 4 : 3 - (int)b;
}

// The head of an EMPTY bucket is NIL: set when a read is taken (here) and when a pop empties a bucket (nt_pop), so that a push
// links `next = head` without asking the bitmap whether the bucket holds anything (nine tests and selects less per expansion)
PS_HD void nt_heads_init(BtMem &m, int n_buckets)
{
    uint32_t *hw = reinterpret_cast<uint32_t *>(m.heads16);
    for (int p = 0; p < (n_buckets + 1) >> 1; ++p) hw[p] = 0xFFFFFFFFu;
}
// The estimate of a read's best score (nt_tail) has failed -- the first hit is worse, or the stack ran empty without a hit: the
// search starts over from the root without one.  The read is still in the lane's local memory (no hit has been recorded, so
// nothing has touched its bounds), the budget is still the read's own; only the search state is reset, on the stack the lane holds.
PS_COLD void nt_restart_without_estimate(const BtArgs &a, NLane &L, BtMem &m)
{
    L.ctl = (L.ctl & NL_BIG) | (uint32_t)M_POP | NL_HAVE_CUR;       // status RS_OK, no hit
    L.kr = 0; L.lr = (uint32_t)a.ix.seq_len; L.wa = (uint32_t)nl_len(L); L.wb = NW_ROOT_C << 6;
    L.lim = (L.lim & 0x00FFFF00u) | 0xffu;                          // no best score, none counted; budget and length stay
    L.nsb = 0; L.bm0 = L.bm1 = 0; L.fh = 0xffffu; L.n_phantom = 0xFF000000u;
    nt_heads_init(m, a.md.n_buckets);
}
PS_HD void nt_finish_read(const BtArgs &a, NLane &L)
{
    a.n_aln[L.r] = nl_n_aln(L.ctl);
    a.status[L.r] = (uint8_t)nl_status(L.ctl);
    L.ctl = nl_set_mode(L.ctl, M_FETCH);
}

// a hit: the current entry reached i == 0 (upstream bwt_match_gap's hit block; same steps as bt_hit)
PS_COLD void nt_hit(const BtArgs &a, NLane &L, BtMem &m)
{
    const Model &md = a.md;
    const uint32_t c = nw_c(L.wb);
    if (a.ix.jump && c != NW_ROOT_C) {           // an entry of the jump table's levels holds (string index, width): its interval is in the parent's slot
        const int d = nt_depth(nl_len(L), L.wa, L.wb);
        if (d < a.ix.jump_levels) {
            const uint32_t *ps = a.ix.jump + (size_t)(jump_level_off(d - 1) + (L.kr >> 2)) * PS_JUMP_SLOT_WORDS;
            L.kr = ps[2 * c] + 1u; L.lr = ps[2 * c + 1];
        }
    }
    const bwtint base = c == NW_ROOT_C ? (a.ix.seq_len & ~(bwtint)0xFFFFFFFFull) : a.ix.L2[c];
    const bwtint k = base + L.kr, l = base + L.lr;
    const int score = nw_score(L.wb), n_gapo = nw_gapo(L.wa);
    const int units = md.profile ? score : nw_mm(L.wa) + n_gapo + (md.mode_gape ? nw_gape(L.wa) : 0);
    const int n_aln = nl_n_aln(L.ctl);
    if (n_aln == 0) {
        if (units > nl_est(L)) { nt_restart_without_estimate(a, L, m); return; }      // the best hit is worse than the estimate children were left out by (nt_tail)
        int t = units + md.u_tight;
        const int own = nl_max_units(L);             // still the read's own budget: nothing has tightened it before the first hit
        t = t > own ? own : t;
        L.lim = (L.lim & 0x00FF0000u) | (uint32_t)score | ((uint32_t)t << 8);      // no best hit counted yet
    }
    if (score == nl_best_score(L)) {             // best_cnt is only ever compared with max_top2 (< 255 on the narrow tiers): kept saturating
        const unsigned long long s = (unsigned long long)(L.lim >> 24) + (unsigned long long)(l - k) + 1ull;
        L.lim = (L.lim & 0x00FFFFFFu) | ((s > 255ull ? 255u : (uint32_t)s) << 24);
    } else if ((int)(L.lim >> 24) > md.max_top2) { nt_finish_read(a, L); return; }
    AlnRec *out = a.alns + (size_t)L.r * a.aln_cap;
    if (n_gapo) {
        for (int j = 0; j < n_aln; ++j)
            if (out[j].k == k && out[j].l == l) return;
    }
    // shadow: discount this hit's occurrences from the width bounds left of the last difference
    {
        const uint32_t shadow = a.ix.seq_len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a.ix.seq_len;   // marks stay above every real width
        uint32_t x = (uint32_t)(l - k + 1), j = 0, prev = 0;
        const int lim = nw_ldp(L.wa);
        for (int i = 0; i < lim; ++i) {
            size_t off = (size_t)i * a.n_reads + L.r;
            uint32_t w = a.w[off];
            int bid = m.cw[i] & 0x7f;
            if (w > x) { w -= x; a.w[off] = w; }
            else if (w == x) {
                // a bound that drops: children skipped on the strength of it (n_phantom) might have been viable, the
                // read is redone by the wide tier, which skips nothing (see bt_hit)
                if (bid > 1 && nl_phantom(L)) L.ctl = nl_set_status(L.ctl, RS_OVERFLOW_POOL);
                bid = 1; w = shadow - (++j); a.w[off] = w;
            }
            m.cw[i] = cw_pack(bid, i > 0 && w == prev);
            prev = w;
        }
        if (lim > 0) {
            uint32_t w = a.w[(size_t)lim * a.n_reads + L.r];
            m.cw[lim] = cw_pack(m.cw[lim] & 0x7f, w == prev);
        }
    }
    if (n_aln >= a.aln_cap) { L.ctl = nl_set_status(L.ctl, RS_OVERFLOW_ALN); nt_finish_read(a, L); return; }
    AlnRec rec;
    rec.k = k; rec.l = l; rec.score = (uint16_t)score; rec.units = (uint16_t)units;
    rec.n_mm = (uint8_t)nw_mm(L.wa); rec.n_gapo = (uint8_t)n_gapo; rec.n_gape = (uint8_t)nw_gape(L.wa);
    rec.n_ins = (uint8_t)nw_ins(L.wb); rec.n_del = (uint8_t)nw_del(L.wb);
    for (int j = 0; j < 7; ++j) rec.pad[j] = 0;
    out[n_aln] = rec;
    L.ctl += 1u << 16;
}

// load one read into the lane's local memory and reset the search state; false if the read is rejected outright
PS_COLD bool nt_fetch(const BtArgs &a, NLane &L, BtMem &m, int q)
{
    const Model &md = a.md;
    const int r = a.order ? a.order[q] : q;            // queue position -> read (the launch's hand-out order)
    const int len = a.lens ? a.lens[r] : a.len;
    L.r = r;
    L.ctl = (uint32_t)M_FETCH;                           // status RS_OK, no hit, no current entry, on the private stack slice
    L.n_phantom = 0xFF000000u;                           // nothing counted, no estimate yet (set below, behind the outright rejection)
    // the budget is the read's own (a launch holds reads of every length that shares the seed rule; md is the longest read's)
    const int max_units = (a.lens && a.units_by_len) ? (int)a.units_by_len[len] : md.max_units;
    L.lim = 0xffu | ((uint32_t)max_units << 8) | ((uint32_t)len << 16);
    {
        const int ncw = lm_ncw(len), ncsw = lm_ncsw(md.seed_len);
        uint32_t *cw32 = reinterpret_cast<uint32_t *>(m.cw), *csw32 = reinterpret_cast<uint32_t *>(m.csw);
        for (int p = 0; p < ncw; ++p) cw32[p] = a.cwb[(size_t)p * a.n_reads + r];
        for (int p = 0; p < ncsw; ++p) csw32[p] = a.cswb[(size_t)p * a.n_reads + r];
        const int nbw = (len + 15) >> 4;
        for (int p = 0; p < nbw; ++p) m.rb[p] = a.bases[(size_t)p * a.n_reads + r];
    }
    int nNu = 0;
    L.rn0 = L.rn1 = 0;
    const int nmw = (len + 31) >> 5;
    for (int p = 0; p < nmw; ++p) {
        uint32_t w = a.nmask[(size_t)p * a.n_reads + r];
        if (lm_nmask_in_regs(a.len)) { if (p == 0) L.rn0 = w; else L.rn1 = w; }
        else m.rn[p] = w;
        nNu += (int)ps_popc(w) * (int)(md.u_mm_pk[4] & 0xffu);
    }
    if (nNu > max_units) { nt_finish_read(a, L); return false; }
    L.kr = 0; L.lr = (uint32_t)a.ix.seq_len; L.wa = (uint32_t)len; L.wb = NW_ROOT_C << 6;     // the root: i = len, state M, score 0
    L.ctl |= NL_HAVE_CUR;
    L.nsb = 0; L.bm0 = L.bm1 = 0; L.fh = 0xffffu;
    {   // the estimate counts only where it is tighter than the read's own budget (else 255: none -- a read without a hit then needs no second search)
        int e = 255;
        if (a.cap_est) { e = (int)a.est[r] - (a.cap_est - 1); if (e < 0) e = 0; }       // cap_est - 1: subtracted from the estimate (tests: makes it fail, PS_CAP_BIAS)
        L.n_phantom = (uint32_t)(e + md.u_tight < max_units ? e : 255) << 24;
    }
    nt_heads_init(m, md.n_buckets);
    return true;
}

// ---- pushes and pops -------------------------------------------------------------------------------------------
// The caller has checked once per expansion that nine free slots remain, and the narrow tiers have fewer than 64 score
// buckets.  A slot: the one popped last first, then fresh ones.
PS_HD uint32_t nt_slot(NLane &L)
{
    const uint32_t fr = L.fh;
    const bool reuse = fr != 0xffffu;
    const uint32_t idx = reuse ? fr : (L.nsb >> 16);
    L.nsb += reuse ? 1u : 0x10001u;              // one more live entry; a fresh slot also moves the bump mark
    L.fh = 0xffffu;
    return idx;
}
// one entry (tests; the expansion below stores its children in two passes)
PS_HD void nt_push(NLane &L, BtMem &m, uint32_t kr, uint32_t lr, uint32_t wa, uint32_t wb)
{
    const int score = nw_score(wb);
    const uint32_t idx = nt_slot(L);
    const uint32_t next = (uint32_t)m.heads16[score];        // NIL while the bucket is empty (nt_heads_init, nt_pop)
    Entry16 e; e.k = kr; e.l = lr; e.a = wa; e.b = wb | (next << 16);
    store16(reinterpret_cast<Entry16 *>(m.pool) + idx, e);
    m.heads16[score] = (uint16_t)idx;
    if (score < 32) L.bm0 |= 1u << score; else L.bm1 |= 1u << (score - 32);
}

// pop the newest entry of the lowest non-empty bucket into the lane's current-entry registers.  NB32: the cost model has at
// most 32 score buckets (every default one has): the bitmap of non-empty buckets is handled as ONE word -- the 64-bit shifts,
// tests and ors of the general form run at a fraction of the 32-bit rate, nine times per expansion
template <bool NB32>
PS_HD void nt_pop(NLane &L, BtMem &m)
{
#ifdef __HIP_DEVICE_COMPILE__
    const int b = (NB32 || L.bm0) ? __ffs((int)L.bm0) - 1 : 32 + __ffs((int)L.bm1) - 1;
#else
    const int b = (NB32 || L.bm0) ? __builtin_ctz(L.bm0) : 32 + __builtin_ctz(L.bm1);
#endif
    const uint32_t hd = m.heads16[b];
    Entry16 e;
    load16(reinterpret_cast<const Entry16 *>(m.pool) + hd, e);
    const uint32_t next = e.b >> 16;
    L.fh = hd;
    m.heads16[b] = (uint16_t)next;                           // NIL when this was the bucket's last entry
    if (next == PS_NIL16) { if (NB32 || b < 32) L.bm0 &= ~(1u << b); else L.bm1 &= ~(1u << (b - 32)); }
    L.kr = e.k; L.lr = e.l; L.wa = e.a; L.wb = e.b & 0xffffu;
    L.nsb -= 1u;
}

// One iteration of a lane (the narrow tiers' bt_iter).  fetch_r: the read this lane may take if it is idle
// (M_FETCH): < 0 = none offered now, >= n_reads = the input is exhausted (retire), else the read index.
// serve_hit: lanes that reached a hit record it now.  big_cap: capacity of a large stack slot.
// Three parts, each entered by the whole wave: nt_head (hits, new reads, the pop and its checks; returns M_EXACT /
// M_EXPAND for the lanes that go on, 0 for the others), nt_step_occ (the memory step), nt_tail (exact extension or
// expansion and pushes).
template <bool STATS, bool NB32>
PS_HD int nt_head(const BtArgs &a, const BtHot &h, NLane &L, LaneStats &st, BtMem &m, int fetch_r, bool serve_hit)
{
    int mode = nl_mode(L.ctl);
    if (mode == M_EXIT || mode == M_GROW) return 0;
    if (STATS) ++st.iters;
    if (mode == M_HIT) {
        if (!serve_hit) return 0;
        L.ctl = nl_set_mode(L.ctl, M_POP);
        { NLane t = L; nt_hit(a, t, m); L = t; }       // may finish the read (mode becomes M_FETCH); by-value round trip: only the copy is address-taken
        return 0;
    }
    if (mode == M_FETCH) {
        if (fetch_r < 0) return 0;
        if (fetch_r >= (int)h.n_reads) { L.ctl = nl_set_mode(L.ctl, M_EXIT); return 0; }
        { NLane t = L; const bool ok = nt_fetch(a, t, m, fetch_r); L = t; if (!ok) return 0; }
        L.ctl = nl_set_mode(L.ctl, M_POP);
        mode = M_POP;
    }
    if (mode == M_POP) {
        const int n_virtual = nl_n_stack(L) + ((L.ctl & NL_HAVE_CUR) ? 1 : 0);
        if (nl_phantom(L) && (long long)n_virtual + (long long)nl_phantom(L) > (long long)h.max_entries) {
            // with the skipped children counted the stack-size stop rule might have fired: the exact count is only
            // kept by the wide tier, which stores every child
            L.ctl = nl_set_status(L.ctl, RS_OVERFLOW_POOL); nt_finish_read(a, L); return 0;
        }
        if (n_virtual == 0 && nl_est(L) != 255 && nl_n_aln(L.ctl) == 0 && nl_status(L.ctl) == RS_OK) { NLane t = L; nt_restart_without_estimate(a, t, m); L = t; return 0; }   // no hit where the estimate promised one
        if (n_virtual == 0 || n_virtual > (int)h.max_entries || nl_status(L.ctl) != RS_OK) { nt_finish_read(a, L); return 0; }
        if (L.ctl & NL_HAVE_CUR) L.ctl &= ~NL_HAVE_CUR;
        else { nt_pop<NB32>(L, m); if (STATS) ++st.pops; }
        L.ctl = nl_set_mode(L.ctl, M_POP);
        const int score = nw_score(L.wb), i1 = nw_i(L.wa);
        if (score > nl_best_score(L) + h.s_stop()) { nt_finish_read(a, L); return 0; }
        const int rem = nl_max_units(L) - nt_units(h, L.wa, L.wb);
        if (rem < 0) return 0;
        const int mleft = (int)(ps_mul24((uint32_t)rem, h.inv_c_min) >> 16);     // rem / c_min
        if (i1 > 0 && mleft < (int)(m.cw[i1 - 1] & 0x7f)) return 0;
        if (i1 == 0) { L.ctl = nl_set_mode(L.ctl, M_HIT); return 0; }
        const int st_ = nw_state(L.wa);
        mode = (mleft == 0 && (st_ == ST_M || h.mode_gape() || nw_gape(L.wa) == h.max_gape())) ? M_EXACT : M_EXPAND;
    }
    return mode;      // M_EXACT or M_EXPAND (M_EXPAND also straight from a stack move, M_GROW)
}

// what a lane carries from the memory step to the tail
struct NtStep { uint32_t ck[4], cl[4]; uint32_t cw_i, cw_im1, cs_i, cs_im1; int s, i; bool in_seed, tm, ctm; };     // tm / ctm: the entry / its children are in the jump table's levels

template <bool STATS>
PS_HD void nt_step_occ(const BtHot &h, NLane &L, LaneStats &st, BtMem &m, NtStep &q)
{
    const int len = nl_len(L), max_len = h.len();      // the read's own length; the launch's longest (layout of the local memory)
    const int i = nw_i(L.wa) - 1;
    // What the step needs from local memory is asked for before the Occ blocks, so that it arrives under their latency:
    // the D bounds of positions i-1 and i, the seed bounds, the read base.
    q.i = i;
    q.cw_i = m.cw[i]; q.cw_im1 = m.cw[i > 0 ? i - 1 : 0];
    const int ii = i - (len - h.seed_len());
    q.in_seed = h.use_seed() && len > h.seed_len() && ii > 0;      // the read's own seed rule (upstream: a read no longer than the seed has none)
    q.cs_i = q.in_seed ? m.csw[ii] : 0u; q.cs_im1 = q.in_seed ? m.csw[ii - 1] : 0u;
    q.s = nt_seq_at(m, L, i, len, max_len);
    const int depth = nt_depth(len, L.wa, L.wb);
    q.tm = depth < (int)h.jump_levels; q.ctm = depth + 1 < (int)h.jump_levels;
    // (the Occ branch first in the source: hipcc then lays the table's two loads behind the Occ path and lets them run on
    // into the expansion -- 3 % against the other order, measured; see the note in nt_occ)
    if (!q.tm) nt_occ<STATS>(h, nw_c(L.wb), L.kr, L.lr, q.ck, q.cl, st);
    else nt_jump(h, depth, L.kr, q.ck, q.cl);
}

template <bool STATS, bool NB32>
PS_HD void nt_tail(const BtHot &h, NLane &L, LaneStats &st, BtMem &m, const NtStep &q, int mode)
{
    const int len = nl_len(L), i = q.i, s = q.s;
    const uint32_t cw_i = q.cw_i, cw_im1 = q.cw_im1, cs_i = q.cs_i, cs_im1 = q.cs_im1;
    const bool in_seed = q.in_seed, tm = q.tm, ctm = q.ctm;
    uint32_t ck[4], cl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { ck[c] = q.ck[c]; cl[c] = q.cl[c]; }
    // child of text symbol c: (c, ck[c]+1, cl[c]), non-empty iff ck[c] < cl[c]
    if (mode == M_EXACT) {            // no difference left: extend exactly, one base per iteration (upstream's
        if (STATS) ++st.exact;        // bwt_match_exact_alt: no bound is looked at again until the read's first base)
        L.ctl = nl_set_mode(L.ctl, M_POP);
        if (s > 3) return;
        const uint32_t ok = sel4(ck, s), ol = sel4(cl, s);
        if (ok >= ol) return;
        L.kr = ctm ? (L.kr << 2) | (uint32_t)s : ok + 1u; L.lr = ctm ? ol - ok : ol; L.wa -= 1u; L.wb = (L.wb & ~NW_C_MASK) | ((uint32_t)s << 6);     // --i; the rest of the entry stands
        L.ctl = nl_set_mode(L.ctl, i == 0 ? M_HIT : M_EXACT);
        return;
    }
    if (STATS) ++st.nodes;
    const uint32_t cap = (L.ctl & NL_BIG) ? h.big_cap : h.pool_cap;
    if (nl_bump(L) + 9u > cap) {          // stack full: ask for a large slot once; if that is full too, the read goes to the next tier
        if (!(L.ctl & NL_BIG) && h.has_big()) { L.ctl = nl_set_mode(L.ctl, M_GROW); return; }
        // a stack of 65,535 entries (a large slot, or the second tier's private one) is the deepest a 16-bit link reaches: straight to the wide tier
        L.ctl = nl_set_mode(nl_set_status(L.ctl, cap >= 65535u ? RS_OVERFLOW_DEEP : RS_OVERFLOW_POOL), M_POP); return;
    }
    const int max_units = nl_max_units(L);
    const int e_un = nt_units(h, L.wa, L.wb), e_sc = nw_score(L.wb);
    const int rem = max_units - e_un;
    const uint32_t inv = h.inv_c_min;
    const int mleft = (int)(ps_mul24((uint32_t)rem, inv) >> 16);
    const int srem = h.seed_units() - e_un;                  // seed budget, in units like the read budget
    const int m_seed = srem <= 0 ? 0 : (int)(ps_mul24((uint32_t)srem, inv) >> 16);
    bool allow_diff = true, allow_M = true;
    const int bnd_same = i > 0 ? (int)(cw_im1 & 0x7f) : 0;   // D bound a child at position i must still afford
    const int bnd_del = (int)(cw_i & 0x7f);                   // ... and a deletion child (it stays at i+1)
    if (i > 0) {
        if (bnd_same > mleft - 1) allow_diff = false;
        else if (bnd_same == mleft - 1 && bnd_del == mleft - 1 && (cw_i & 0x80u)) allow_M = false;
        if (in_seed) {
            const int s1 = (int)(cs_im1 & 0x7f), s0 = (int)(cs_i & 0x7f);
            if (s1 > m_seed - 1) allow_diff = false;
            else if (s1 == m_seed - 1 && s0 == m_seed - 1 && (cs_i & 0x80u)) allow_M = false;
        }
    }
    const int e_go = nw_gapo(L.wa), e_ge = nw_gape(L.wa), e_st = nw_state(L.wa);
    const int tmp = e_go + e_ge;
    const bool gap_ok = allow_diff && i >= h.indel_end_skip() + tmp && len - i >= h.indel_end_skip() + tmp;
    const bool from_m = e_st == ST_M, from_i = e_st == ST_I, from_d = e_st == ST_D;
    const uint32_t wa_keep = L.wa & NW_KEEP, pos2 = (uint32_t)i | ((uint32_t)i << 8);       // children are differences: last_diff_pos = i
    const uint32_t wb0 = L.wb & 0x1ffu;                                         // n_ins, n_del, c (the score is set per child)
    const uint32_t pkr = L.kr, plr = L.lr;
    // A child that costs c units is a candidate iff c <= rem (it fits the budget).  It is popped only to be dropped when the budget
    // left after it cannot pay for the differences its remaining bases need at least (the check every pop starts with) --
    // floor((rem - c) / c_min) < bound, i.e. c > rem - bound * c_min.  The budget only ever shrinks, so such a child is dropped
    // whenever it is popped: it is not stored at all, only counted (n_phantom) for the stack-size stop rule.  In profile mode this
    // is about half of all pops.  One threshold per bound, two compares per child.
    //   The estimate of the read's best score (ps_effort.hip; in units, profile costs only) tightens that before the first hit: the
    // budget will then be at most est + u_tight, and a child whose score is above est is not popped before that hit (lower scores go
    // first), so it is dropped when popped if it needs more than est + u_tight -- also left out and counted.  If the estimate turns out
    // too low (first hit worse than est, or no hit) the search of the read starts over without one (nt_restart_without_estimate): results never depend on it.
    const int need_same = (int)ps_mul24((uint32_t)bnd_same, h.c_min), need_del = (int)ps_mul24((uint32_t)bnd_del, h.c_min);
    const int cap_rem = nl_est(L) - e_sc, ut = h.u_tight();
    const int cap_same = cap_rem + (ut > need_same ? ut - need_same : 0), cap_del = cap_rem + (ut > need_del ? ut - need_del : 0);
    const int thr_same = (rem - need_same) < cap_same ? (rem - need_same) : cap_same, thr_del = (rem - need_del) < cap_del ? (rem - need_del) : cap_del;
    uint32_t phantom = 0, gmask = 0;
    // ---- which children are stored, and their scores (push order: insertion, deletion of A C G T, mismatches) ----
    int sc[9];
    uint32_t wa_i, wa_d;
    {   // insertion child: opens from M, extends from I; keeps the parent's interval
        const bool open = from_m && e_go < h.max_gapo(), ext = from_i && e_ge < h.max_gape();
        const int c = open ? h.u_gapo_ins() : h.u_gape();
        sc[0] = e_sc + (open ? h.s_gapo_ins() : h.s_gape());
        const bool cand = gap_ok && (open || ext) && c <= rem, go = cand && c <= thr_same;
        gmask |= go ? 1u : 0u;
        phantom += (cand && !go) ? 1u : 0u;
        wa_i = wa_keep + pos2 + ((uint32_t)ST_I << 24) + (open ? 1u << 26 : 1u << 29);
    }
    {   // deletion children: open from M, extend from D; the four share score, counts and position (they stay at i+1)
        const uint32_t occ = tm ? plr : plr - pkr + 1u;    // never the root here (from_d)
        const bool open = from_m && e_go < h.max_gapo();
        const bool ext = from_d && e_ge < h.max_gape() && ((e_ge + e_go) * h.u_tight() < max_units || occ < (uint32_t)h.max_del_occ());
        const int c = open ? h.u_gapo_del() : h.u_gape();
        const int scd = e_sc + (open ? h.s_gapo_del() : h.s_gape());
        const bool cand = gap_ok && (open || ext) && c <= rem, go = cand && c <= thr_del;
        const uint32_t ne = (ck[0] < cl[0] ? 2u : 0u) | (ck[1] < cl[1] ? 4u : 0u) | (ck[2] < cl[2] ? 8u : 0u) | (ck[3] < cl[3] ? 16u : 0u);
        gmask |= go ? ne : 0u;
        if (cand && !go) phantom += ps_popc(ne);
        sc[1] = sc[2] = sc[3] = sc[4] = scd;
        wa_d = wa_keep + ((uint32_t)(i + 1) | ((uint32_t)(i + 1) << 8)) + ((uint32_t)ST_D << 24) + (open ? 1u << 26 : 1u << 29);
    }
    const bool do_mm = allow_diff && allow_M;
    const uint32_t s_word = cost_word(h.s_pk, s), u_word = h.profile() ? s_word : cost_word(h.u_pk, s);   // this read base against each text symbol
    uint32_t xk[4], xl[4];             // the four symbols in push order (read base + 1, + 2, + 3, then the read base itself)
#pragma unroll
    for (int j = 1; j <= 4; ++j) {
        const int c = (s + j) & 3;
        xk[j - 1] = sel4(ck, c); xl[j - 1] = sel4(cl, c);
        sc[4 + j] = 0;
        if (j < 4 || s > 3) {          // mismatch children (the fourth only for an N in the read)
            const int cu = (int)((u_word >> (8 * c)) & 0xffu);
            sc[4 + j] = e_sc + (int)((s_word >> (8 * c)) & 0xffu);
            const bool cand = do_mm && xk[j - 1] < xl[j - 1] && cu <= rem, go = cand && cu <= thr_same;
            gmask |= go ? 1u << (4 + j) : 0u;
            phantom += (cand && !go) ? 1u : 0u;
        }
    }
    // ---- slots: the one popped last first, then fresh ones, all at once (rank of a child among the stored ones) ----
    const uint32_t n_push = ps_popc(gmask), fr = L.fh;
    const bool reuse = fr != 0xffffu && n_push != 0u;
    const uint32_t slot0 = (L.nsb >> 16) - (reuse ? 1u : 0u);
    L.nsb += n_push + ((n_push - (reuse ? 1u : 0u)) << 16);
    L.fh = n_push ? 0xffffu : fr;
    if (STATS) st.pushes += n_push;
    // ---- pass 1: bucket heads.  Local memory executes a wave's accesses in order, so every head read below sees the head
    // writes before it; nothing is waited for until all of them are under way.  (An empty bucket's head is NIL: nt_heads_init.) ----
    uint32_t idx[9], raw[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        idx[j] = 0; raw[j] = 0;
        if ((gmask >> j) & 1u) {
            const uint32_t rank = ps_popc(gmask & ((1u << j) - 1u));
            idx[j] = (reuse && rank == 0u) ? fr : slot0 + rank;
            raw[j] = m.heads16[sc[j]];
            m.heads16[sc[j]] = (uint16_t)idx[j];
            if (NB32 || sc[j] < 32) L.bm0 |= 1u << sc[j]; else L.bm1 |= 1u << (sc[j] - 32);
        }
    }
#ifdef __HIP_DEVICE_COMPILE__
    // the heads read above are first looked at here (one wait for all of them): without this fence the compiler
    // shifts each into place right behind its read and waits nine times
    asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]), "+v"(raw[8]));
#endif
    // ---- pass 2: the entries ----
    Entry16 *const pool = reinterpret_cast<Entry16 *>(m.pool);
#define NT_NEXT(j) (raw[j] << 16)
    if (gmask & 1u) { Entry16 e; e.k = pkr; e.l = plr; e.a = wa_i; e.b = (wb0 + 1u) | ((uint32_t)sc[0] << 9) | NT_NEXT(0); store16(pool + idx[0], e); }
    {
        const uint32_t wb_d = ((wb0 & 0x3fu) + (1u << 3)) | ((uint32_t)sc[1] << 9);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if ((gmask >> (1 + c)) & 1u) { Entry16 e; e.k = ctm ? (pkr << 2) | (uint32_t)c : ck[c] + 1u; e.l = ctm ? cl[c] - ck[c] : cl[c]; e.a = wa_d; e.b = wb_d | ((uint32_t)c << 6) | NT_NEXT(1 + c); store16(pool + idx[1 + c], e); }
    }
    {
        const uint32_t wa_x = wa_keep + pos2 + (1u << 16), wb_x = wb0 & 0x3fu;
#pragma unroll
        for (int j = 1; j <= 4; ++j) {
            const uint32_t c = (uint32_t)((s + j) & 3);
            if ((gmask >> (4 + j)) & 1u) { Entry16 e; e.k = ctm ? (pkr << 2) | c : xk[j - 1] + 1u; e.l = ctm ? xl[j - 1] - xk[j - 1] : xl[j - 1]; e.a = wa_x; e.b = wb_x | (c << 6) | ((uint32_t)sc[4 + j] << 9) | NT_NEXT(4 + j); store16(pool + idx[4 + j], e); }
        }
    }
#undef NT_NEXT
    L.ctl = nl_set_mode(L.ctl, M_POP);
    if (s < 4 && xk[3] < xl[3]) {      // the match child: parent's score, pushed last => the next pop: it stays in registers
        L.kr = ctm ? (pkr << 2) | (uint32_t)s : xk[3] + 1u; L.lr = ctm ? xl[3] - xk[3] : xl[3]; L.wa = wa_keep + (uint32_t)i; L.wb = (L.wb & ~NW_C_MASK) | ((uint32_t)s << 6);
        L.ctl |= NL_HAVE_CUR;
    }
    L.n_phantom += phantom;
}

template <bool STATS, bool NB32>
PS_HD void nt_iter(const BtArgs &a, const BtHot &h, NLane &L, LaneStats &st, BtMem &m, int fetch_r, bool serve_hit, NtClock *clk = nullptr)
{
    PS_USTAMP(clk, 0);                                   // read hand-out, stack moves (the kernel loop), up to here
    const int mode = nt_head<STATS, NB32>(a, h, L, st, m, fetch_r, serve_hit);
    PS_USTAMP(clk, 1);                                   // hits, new reads, the pop and its checks
    NtStep q;
    if (mode) nt_step_occ<STATS>(h, L, st, m, q);
#if defined(PS_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (STATS && clk && mode) asm volatile("" :: "v"(q.ck[0]), "v"(q.cl[0]), "v"(q.ck[3]), "v"(q.cl[3]));
#endif
    PS_USTAMP(clk, 2);                                   // the memory step
    if (mode) nt_tail<STATS, NB32>(h, L, st, m, q, mode);
    PS_USTAMP(clk, 3);                                   // exact extension / expansion and pushes
}

}  // namespace ps
