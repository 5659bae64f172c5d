// ps_core.h -- per-lane device logic of the mapping hot path (gfx950 kernels
// in ps_kernels.hip wrap these; tests/hostsim runs the same functions lane by
// lane on the host as a debugging aid -- it is not a product path).
//
// Algorithms: FM-index backward search with bounded best-first backtracking
// (what `bwa aln` / `bwa parasuite` compute for
// /root/reference/src/src/mapping/PARAsuiteMapping.java:63-77 and
// BWAMapping.java:51-61), suffix-array interval -> text position and banded
// global alignment of gapped hits (`bwa samse`, PARAsuiteMapping.java:85-92).
// One read per lane; every loop iteration of a lane performs at most one
// Occ-block pair lookup so that all 64 lanes of a wave keep their random
// 64-byte HBM loads in flight together.
#pragma once
#include "ps_types.h"

namespace ps {

// ------------------------------------------------------------ popcounts ---
PS_HD uint32_t ps_popc(uint32_t x)
{
#ifdef __HIP_DEVICE_COMPILE__
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}
PS_HD uint32_t pfx_mask32(int ns) // mask of the first ns (<=32) symbols of a 32-symbol plane word
{
    return ns >= 32 ? 0xFFFFFFFFu : (ns <= 0 ? 0u : ((1u << ns) - 1u));
}

struct Blk { uint32_t x[16]; };  // cnt[4] + lo[6] + hi[6]

// The kernels take their arguments from a struct in device memory, so the compiler sees generic pointers and
// would emit FLAT loads/stores (which also tie up the LDS counter).  The hot accesses state the address space.
#ifdef __HIP_DEVICE_COMPILE__
#define PS_AS_GLOBAL(T, p) ((const __attribute__((address_space(1))) T *)(p))
#define PS_AS_GLOBAL_W(T, p) ((__attribute__((address_space(1))) T *)(p))
typedef uint32_t ps_u32x4 __attribute__((ext_vector_type(4)));
#endif

PS_HD void load_blk(const OccBlock *blocks, uint32_t b, Blk &o)
{
#ifdef __HIP_DEVICE_COMPILE__
    const __attribute__((address_space(1))) ps_u32x4 *p = PS_AS_GLOBAL(ps_u32x4, blocks + b);
    const ps_u32x4 a = p[0], c = p[1], d = p[2], e = p[3];
    o.x[0] = a.x; o.x[1] = a.y; o.x[2] = a.z; o.x[3] = a.w;
    o.x[4] = c.x; o.x[5] = c.y; o.x[6] = c.z; o.x[7] = c.w;
    o.x[8] = d.x; o.x[9] = d.y; o.x[10] = d.z; o.x[11] = d.w;
    o.x[12] = e.x; o.x[13] = e.y; o.x[14] = e.z; o.x[15] = e.w;
#else
    const uint32_t *p = reinterpret_cast<const uint32_t *>(blocks + b);
    for (int j = 0; j < 16; ++j) o.x[j] = p[j];
#endif
}
// occurrences of every symbol among the first r (1..192) symbols of the block, plus the block base.
// The symbols are stored as two bit planes (lo = x[4..9], hi = x[10..15], 32 symbols per word), so one
// prefix mask and three popcounts per 32 symbols give all four counts.
PS_HD void blk_count4(const Blk &b, int r, uint32_t cnt[4])
{
    uint32_t c1 = 0, c2 = 0, c3 = 0;
    const int q = r >> 5;                                   // whole words before the partial one
    const uint32_t part = (1u << (r & 31)) - 1u;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const uint32_t m = j < q ? 0xFFFFFFFFu : (j == q ? part : 0u), lo = b.x[4 + j] & m, hi = b.x[10 + j] & m;
        c3 += ps_popc(lo & hi);
        c2 += ps_popc(hi & ~lo);
        c1 += ps_popc(lo & ~hi);
    }
    cnt[0] = b.x[0] + ((uint32_t)r - c1 - c2 - c3);
    cnt[1] = b.x[1] + c1; cnt[2] = b.x[2] + c2; cnt[3] = b.x[3] + c3;
}
PS_HD uint32_t blk_count1(const Blk &b, int r, int c)
{
    uint32_t n = 0;
    const uint32_t flo = (c & 1) ? 0u : 0xFFFFFFFFu, fhi = (c & 2) ? 0u : 0xFFFFFFFFu;   // XOR masks: symbol c reads 1,1
#pragma unroll
    for (int j = 0; j < 6; ++j) n += ps_popc((b.x[4 + j] ^ flo) & (b.x[10 + j] ^ fhi) & pfx_mask32(r - 32 * j));
    uint32_t base = c == 0 ? b.x[0] : (c == 1 ? b.x[1] : (c == 2 ? b.x[2] : b.x[3]));
    return base + n;
}
PS_HD int blk_sym(const Blk &b, int pos) // symbol pos (0..191) of the block
{
    uint32_t lo = 0, hi = 0; const int wi = pos >> 5;
#pragma unroll
    for (int j = 0; j < 6; ++j) { lo = (j == wi) ? b.x[4 + j] : lo; hi = (j == wi) ? b.x[10 + j] : hi; }
    return (int)(((lo >> (pos & 31)) & 1u) | (((hi >> (pos & 31)) & 1u) << 1));
}

// build one block from up to 192 symbols (codes 0..3) and the running counts before it
PS_HD void blk_pack(OccBlock &b, const uint8_t *syms, int n, const uint32_t cnt[4])
{
    for (int c = 0; c < 4; ++c) b.cnt[c] = cnt[c];
    for (int j = 0; j < 6; ++j) {
        uint32_t lo = 0, hi = 0;
        for (int t = 0; t < 32; ++t) { int p = j * 32 + t; if (p < n) { lo |= (uint32_t)(syms[p] & 1) << t; hi |= (uint32_t)((syms[p] >> 1) & 1) << t; } }
        b.sym[j] = lo; b.sym[6 + j] = hi;
    }
}

// register-only selects (dynamic indexing of small arrays would push them to scratch memory)
// two-level selects on the bits of c: three conditional moves, no branches and nothing the compiler can
// turn back into an indexed (scratch) array
PS_HD uint32_t sel4s(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, int c)
{
    const bool b0 = (c & 1) != 0, b1 = (c & 2) != 0;
    const uint32_t lo = b0 ? v1 : v0, hi = b0 ? v3 : v2;
    return b1 ? hi : lo;
}
PS_HD uint32_t sel4(const uint32_t v[4], int c) { return sel4s(v[0], v[1], v[2], v[3], c); }
PS_HD bwtint L2_of(const IndexView &ix, int c)
{
    const bool b0 = (c & 1) != 0, b1 = (c & 2) != 0;
    const bwtint lo = b0 ? ix.L2[1] : ix.L2[0], hi = b0 ? ix.L2[3] : ix.L2[2];
    return b1 ? hi : lo;
}
// cost of reading text symbol c where the (reverse-complemented) read has s: four byte lanes per read symbol
PS_HD uint32_t cost_word(const uint32_t pk[5], int s)        // the four costs of read symbol s, one per byte
{
    const uint32_t w = sel4s(pk[0], pk[1], pk[2], pk[3], s);
    return s >= 4 ? pk[4] : w;
}
PS_HD int cost_of(const uint32_t pk[5], int s, int c) { return (int)((cost_word(pk, s) >> (8 * c)) & 0xffu); }

// row (0..n) of the BW matrix of T$ -> index into the stored BWT (the '$' row is not stored)
PS_HD bwtint row_to_stored(bwtint primary, bwtint row) { return row - (row >= primary ? 1u : 0u); }
// stored symbol index -> its 192-symbol block and the offset inside it (32-bit arithmetic: s < 2^33)
PS_HD uint32_t blk_of(bwtint s, int &off)
{
    const uint32_t b = (uint32_t)(s >> 6) / 3u;
    off = (int)((uint32_t)s - b * (uint32_t)PS_BLK_SYMS);
    return b;
}
// interval size as the 32-bit quantity the width arrays hold: only the full range [0, n] can exceed 32 bits
PS_HD uint32_t width32(bwtint k, bwtint l) { const bwtint w = l - k + 1; return w > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)w; }

struct LaneStats { uint32_t pairs, same, nodes, pushes, pops, iters, exact, lf; };

// Occ(k-1, .) and Occ(l, .) for all four symbols: the memory operation of one search step.
PS_HD void occ_pair4(const OccBlock *blocks, bwtint primary, bwtint k, bwtint l, uint32_t ck[4], uint32_t cl[4], LaneStats &st)
{
    int ol_ = 0, ok_ = 0;
    const uint32_t bl = blk_of(row_to_stored(primary, l), ol_);
    uint32_t bk = bl;
    bool need_k = k != 0;
    if (need_k) bk = blk_of(row_to_stored(primary, k - 1), ok_);
    Blk xl, xk;
    load_blk(blocks, bl, xl);
    bool other = need_k && bk != bl;
    if (other) load_blk(blocks, bk, xk);
    blk_count4(xl, ol_ + 1, cl);
    // one counting pass for k-1 whichever block it is in (two passes under complementary lane masks would both
    // be executed by a wave that has lanes of either kind)
#pragma unroll
    for (int j = 0; j < 16; ++j) xk.x[j] = other ? xk.x[j] : xl.x[j];
    blk_count4(xk, need_k ? ok_ + 1 : 0, ck);
    if (!need_k) { ck[0] = ck[1] = ck[2] = ck[3] = 0; }
    ++st.pairs;
    if (need_k && !other) ++st.same;
}
PS_HD void occ_pair1(const IndexView &ix, bwtint k, bwtint l, int c, uint32_t &ok, uint32_t &ol, LaneStats &st)
{
    int ol_ = 0, ok_ = 0;
    const uint32_t bl = blk_of(row_to_stored(ix.primary, l), ol_);
    uint32_t bk = bl;
    bool need_k = k != 0;
    if (need_k) bk = blk_of(row_to_stored(ix.primary, k - 1), ok_);
    Blk xl, xk;
    load_blk(ix.blocks, bl, xl);
    bool other = need_k && bk != bl;
    if (other) load_blk(ix.blocks, bk, xk);
    ol = blk_count1(xl, ol_ + 1, c);
    if (!need_k) ok = 0;
    else if (other) ok = blk_count1(xk, ok_ + 1, c);
    else ok = blk_count1(xl, ok_ + 1, c);
    ++st.pairs;
    if (need_k && !other) ++st.same;
}

// ------------------------------------------------------------ read access --
PS_HD int read_base(const uint32_t *bases, const uint32_t *nmask, int n_reads, int r, int j) // 0..3, 4 = N
{
    uint32_t w = bases[(size_t)(j >> 4) * n_reads + r];
    uint32_t m = nmask[(size_t)(j >> 5) * n_reads + r];
    return ((m >> (j & 31)) & 1u) ? 4 : (int)((w >> (2 * (j & 15))) & 3u);
}

// compact width byte: low 7 bits = min(bid,127), bit 7 = (w[i-1] == w[i])
PS_HD uint8_t cw_pack(int bid, bool eq) { return (uint8_t)((bid > 127 ? 127 : bid) | (eq ? 0x80 : 0)); }

// --------------------------------------------------------- width chains ---
// Lower bound D(i) on the differences needed for the read's last i+1 bases
// (what upstream's bwt_cal_width returns), computed by restartable exact
// backward search.  One chain = (k,l,bid); the kernel steps the full-read chain
// and the seed chain together so a lane keeps two independent loads in flight.
struct WChain { bwtint k, l; int bid; uint32_t prev_w; };
PS_HD void wchain_init(const IndexView &ix, WChain &c) { c.k = 0; c.l = ix.seq_len; c.bid = 0; c.prev_w = 0xFFFFFFFFu; }
PS_HD void wchain_step(const IndexView &ix, WChain &c, int base, uint32_t &w_out, uint8_t &cw_out, bool first, LaneStats &st)
{
    if (base < 4) {
        uint32_t ok, ol;
        occ_pair1(ix, c.k, c.l, base, ok, ol, st);
        const bwtint l2 = L2_of(ix, base);
        c.k = l2 + ok + 1;
        c.l = l2 + ol;
    }
    if (c.k > c.l || base > 3) { c.k = 0; c.l = ix.seq_len; ++c.bid; }
    uint32_t w = width32(c.k, c.l);
    w_out = w;
    cw_out = cw_pack(c.bid, !first && w == c.prev_w);
    c.prev_w = w;
}

// ----------------------------------------------------- backtracking lane ---
static const int PS_POP_TRIES = 1;   // retrying rejected pops inside one iteration was measured: the whole wave pays for the
                                     // extra pop passes (profile mode +21 % time), so one pop per iteration
enum { M_FETCH = 0, M_POP = 1, M_EXACT = 2, M_EXPAND = 3, M_EXIT = 4, M_HIT = 5,
       M_GROW = 6 };   // the lane's private stack is full: the wave moves it to a large slot (kernel), then the expansion is redone

struct BtLane {
    int r, mode;
    // current entry (held in registers; "virtually" on the stack when have_cur)
    bwtint k, l;
    int i, score, units, n_mm, n_gapo, n_gape, n_ins, n_del, state, ldp;
    bool have_cur;
    int best_score, max_units, n_aln, n_stack, status;
    unsigned long long best_cnt;
    unsigned long long bm0, bm1;  // non-empty score buckets (two scalars: a dynamically indexed array would live in scratch)
    uint32_t bump, free_head, iters0;
    uint32_t rn0, rn1;             // N mask of a read of up to 64 bases (read orientation)
    int len;                       // this read's length (reads of one cost class share a launch; the layout is for the longest)
    uint32_t n_phantom;            // narrow stack: children not stored because the D(i) bound already rules them out (see bt_iter)
    uint32_t cap;                  // capacity of the stack this lane currently uses (private slice, or a large slot after M_GROW)
    LaneStats st;
};

// local per-lane memory (LDS in the kernel), all 4-byte words: compact widths cw (4 positions per word),
// seed widths csw, the read's 2-bit base words and N-mask words, then -- narrow stack only -- the heads of
// the score buckets as uint16 entry indices.  The byte count keeps (stride/4) odd so that lanes reading
// the same offset hit different LDS banks.  Reads of up to 64 bases keep their N mask in two lane registers
// instead: those 8 bytes decide between three and four resident workgroups per CU at 50 bp.
PS_HD int lm_ncw(int len) { return (len + 1 + 3) >> 2; }
PS_HD int lm_ncsw(int seed_len) { return seed_len > 0 ? (seed_len + 1 + 3) >> 2 : 0; }
PS_HD bool lm_nmask_in_regs(int len) { return len <= 64; }
PS_HD int lm_heads_off(int len, int seed_len) { return 4 * (lm_ncw(len) + lm_ncsw(seed_len) + ((len + 15) >> 4) + (lm_nmask_in_regs(len) ? 0 : ((len + 31) >> 5))); }
PS_HD int lm_bytes(int len, int seed_len, int n_buckets, bool wide)
{
    int n = lm_heads_off(len, seed_len) + (wide ? 0 : 2 * n_buckets);
    n = (n + 3) & ~3;
    if (((n >> 2) & 1) == 0) n += 4;
    return n;
}

// What every iteration of the backtracking loop needs from the arguments, as a handful of words the kernel
// keeps in scalar registers for its whole life (BtArgs itself lives in device memory for the rare, non-inlined
// paths; reading it inside the loop costs a scalar-cache round trip per field and turns a dynamic L2[c]
// into a vector load).  Small fields are packed four to a word.
// ---- jump table: the child intervals of the short strings ------------------------------------------------------
// A search step at depth d (text symbols consumed so far) needs Occ(k-1, .) and Occ(l, .) of its interval.  Near the root
// the interval is wide, the two rows lie in different blocks of the BWT (two random 64-byte reads), and there are few distinct
// intervals: 4^d strings at depth d.  The table holds, for every string of fewer than K symbols, exactly what nt_occ returns
// for its interval -- eight words {ck[0], cl[0], .. ck[3], cl[3]} (an empty string's slot is all zero: no child) -- so a step
// at depth < K is ONE 32-byte read, and the upper levels (4^d * 32 bytes: 32 MB at d = 10) come out of L2 / the MALL.  At
// hg19 size K = 14: 16.3 G of the bench's 38 G steps, which cost 31 G block reads without it (profiles/r02_kernel_experiments.txt);
// 2.9 GB of the 288.  While a search entry is at depth < K it carries the string's index in its level (kr) and the width of its
// interval (lr) instead of the interval: the interval itself comes back out of the parent's slot where a hit needs it.
// Level d starts (4^d - 1) / 3 slots into the table; a string's index is its symbols as base-4 digits, the last consumed lowest.
static const int PS_JUMP_SLOT_WORDS = 8, PS_JUMP_MAX_LEVELS = 14;
PS_HD uint32_t jump_level_off(int d) { return 0x55555555u & ((1u << (2 * d)) - 1u); }
PS_HD size_t jump_words(int levels) { return (size_t)jump_level_off(levels) * PS_JUMP_SLOT_WORDS; }
// levels worth having: below them an interval is still wider than a fraction of a block on average
PS_HD int jump_levels_for(bwtint seq_len) { int k = 0; while (k < PS_JUMP_MAX_LEVELS && (seq_len >> (2 * k)) >= 48) ++k; return k; }

struct BtHot {
    const OccBlock *blocks; bwtint primary;
    const uint32_t *jump; uint32_t jump_levels;
    uint32_t L2lo[4], L2hi;          // C array of the FM index: low words, bit 32 of L2[c] at bit c (bit 4: bit 32 of n, the root's base)
    uint32_t s_pk[5], u_pk[5];       // substitution costs, one word per read symbol (byte c = text symbol)
    uint32_t p0, p1, p2, p3;
    uint32_t inv_c_min, max_entries, pool_cap, n_reads, big_cap;
    uint32_t c_min;                  // the cheapest difference, in units
    PS_HD int len() const { return (int)(p0 & 0xffu); }
    PS_HD int seed_len() const { return (int)((p0 >> 8) & 0xffu); }
    PS_HD int indel_end_skip() const { return (int)((p0 >> 16) & 0xffu); }
    PS_HD int max_del_occ() const { return (int)(p0 >> 24); }
    PS_HD int max_gapo() const { return (int)(p1 & 0xfu); }
    PS_HD int max_gape() const { return (int)((p1 >> 4) & 0xfu); }
    PS_HD bool mode_gape() const { return ((p1 >> 8) & 1u) != 0; }
    PS_HD bool use_seed() const { return ((p1 >> 9) & 1u) != 0; }
    PS_HD bool profile() const { return ((p1 >> 10) & 1u) != 0; }
    PS_HD bool has_big() const { return ((p1 >> 11) & 1u) != 0; }
    PS_HD int seed_units() const { return (int)((p1 >> 12) & 0xfffu); }      // max_seed_diff * u_tight
    PS_HD int u_tight() const { return (int)(p1 >> 24); }
    PS_HD int s_gapo_ins() const { return (int)(p2 & 0xffu); }
    PS_HD int s_gape() const { return (int)((p2 >> 8) & 0xffu); }
    PS_HD int s_gapo_del() const { return (int)((p2 >> 16) & 0xffu); }
    PS_HD int s_stop() const { return (int)(p2 >> 24); }
    PS_HD int u_gapo_ins() const { return (int)(p3 & 0xffu); }
    PS_HD int u_gape() const { return (int)((p3 >> 8) & 0xffu); }
    PS_HD int u_gapo_del() const { return (int)((p3 >> 16) & 0xffu); }
    PS_HD int n_buckets() const { return (int)(p3 >> 24); }
    PS_HD bwtint L2(int c) const     // c known at compile time in unrolled loops
    { return (bwtint)L2lo[c] | ((bwtint)((L2hi >> c) & 1u) << 32); }
    PS_HD bwtint L2_dyn(int c) const
    { return (bwtint)sel4s(L2lo[0], L2lo[1], L2lo[2], L2lo[3], c) | ((bwtint)((L2hi >> c) & 1u) << 32); }
};
// false if a field does not fit its packed width (the caller reports the model as unsupported)
inline bool bt_hot_make(const BtArgs &a, BtHot &h)
{
    const Model &md = a.md;
    h.blocks = a.ix.blocks; h.primary = a.ix.primary; h.L2hi = 0;
    h.jump = a.ix.jump; h.jump_levels = a.ix.jump ? (uint32_t)a.ix.jump_levels : 0u;
    for (int c = 0; c < 4; ++c) { h.L2lo[c] = (uint32_t)a.ix.L2[c]; h.L2hi |= (uint32_t)((a.ix.L2[c] >> 32) & 1ull) << c; }
    h.L2hi |= (uint32_t)((a.ix.seq_len >> 32) & 1ull) << 4;
    for (int c = 0; c < 5; ++c) { h.s_pk[c] = md.s_mm_pk[c]; h.u_pk[c] = md.u_mm_pk[c]; }
    const int seed_units = md.max_seed_diff * md.u_tight;
    const bool ok = a.len >= 0 && a.len <= 255 && md.seed_len >= 0 && md.seed_len <= 255 && md.indel_end_skip >= 0 && md.indel_end_skip <= 255 &&
                    md.max_del_occ >= 0 && md.max_del_occ <= 255 && md.max_gapo >= 0 && md.max_gapo <= 15 && md.max_gape >= 0 && md.max_gape <= 15 &&
                    seed_units >= 0 && seed_units <= 4095 && md.u_tight >= 0 && md.u_tight <= 255 && md.s_stop >= 0 && md.s_stop <= 255 &&
                    md.s_gapo_ins >= 0 && md.s_gapo_ins <= 255 && md.s_gape >= 0 && md.s_gape <= 255 && md.s_gapo_del >= 0 && md.s_gapo_del <= 255 &&
                    md.u_gapo_ins >= 0 && md.u_gapo_ins <= 255 && md.u_gape >= 0 && md.u_gape <= 255 && md.u_gapo_del >= 0 && md.u_gapo_del <= 255 &&
                    md.n_buckets >= 0 && md.n_buckets <= 255 && md.max_entries >= 0 && md.inv_c_min >= 0 && md.max_units <= 255 && a.ix.L2[0] == 0;
    h.p0 = (uint32_t)a.len | ((uint32_t)md.seed_len << 8) | ((uint32_t)md.indel_end_skip << 16) | ((uint32_t)md.max_del_occ << 24);
    h.p1 = (uint32_t)md.max_gapo | ((uint32_t)md.max_gape << 4) | ((md.mode_gape ? 1u : 0u) << 8) | ((md.use_seed ? 1u : 0u) << 9) |
           ((md.profile ? 1u : 0u) << 10) | ((a.n_big ? 1u : 0u) << 11) | ((uint32_t)seed_units << 12) | ((uint32_t)md.u_tight << 24);
    h.p2 = (uint32_t)md.s_gapo_ins | ((uint32_t)md.s_gape << 8) | ((uint32_t)md.s_gapo_del << 16) | ((uint32_t)md.s_stop << 24);
    h.p3 = (uint32_t)md.u_gapo_ins | ((uint32_t)md.u_gape << 8) | ((uint32_t)md.u_gapo_del << 16) | ((uint32_t)md.n_buckets << 24);
    h.c_min = (uint32_t)md.c_min;
    h.inv_c_min = (uint32_t)md.inv_c_min; h.max_entries = (uint32_t)md.max_entries; h.pool_cap = a.pool_cap; h.n_reads = (uint32_t)a.n_reads; h.big_cap = a.big_cap;
    return ok;
}

struct BtMem {              // views of one lane's slices
    uint8_t *cw, *csw;      // byte views of the width words
    uint32_t *rb, *rn;      // the read: 2-bit bases, N mask (read orientation)
    uint16_t *heads16;      // narrow: in local memory
    void *pool;             // narrow: Entry16[pool_cap]; wide: Entry[pool_cap]
    uint32_t *heads;        // wide: global, PS_MAX_BUCKETS per lane
};

PS_HD void bt_mem_bind(BtMem &m, uint8_t *mine, int len, int seed_len)
{
    m.cw = mine; m.csw = mine + 4 * lm_ncw(len);
    m.rb = reinterpret_cast<uint32_t *>(m.csw + 4 * lm_ncsw(seed_len));
    m.rn = m.rb + ((len + 15) >> 4);
    m.heads16 = reinterpret_cast<uint16_t *>(mine + lm_heads_off(len, seed_len));
}
// base j of the reverse-complemented read (what the search consumes): 0..3, 4 = N
PS_HD int seq_at(const BtMem &m, const BtLane &L, int j, int len, int max_len)
{
    const int p = len - 1 - j;
    const uint32_t b = (m.rb[p >> 4] >> (2 * (p & 15))) & 3u;
    const uint32_t nw = lm_nmask_in_regs(max_len) ? (p < 32 ? L.rn0 : L.rn1) : m.rn[p >> 5];
    return ((nw >> (p & 31)) & 1u) ? 4 : 3 - (int)b;
}

// The stack of one lane.  NARROW (tiers 1-2; ps_narrow.h): 16-byte entries, bump allocation, bucket heads in LDS --
// one 16-byte global store per push and one 16-byte load per pop.  WIDE (last tier, up to the 2,000,000
// live entries upstream allows; below): 32-byte entries with a free list and heads in global memory.
struct Entry16 { uint32_t k, l, a, b; };   // narrow entry: interval relative to its symbol's rows, packed counts, next link (ps_narrow.h)
static const uint32_t PS_NIL16 = 0xFFFFu;

PS_HD bool bm_test(const BtLane &L, int b) { return (((b & 64) ? L.bm1 : L.bm0) >> (b & 63)) & 1ull; }
PS_HD void bm_set(BtLane &L, int b) { const unsigned long long v = 1ull << (b & 63); if (b & 64) L.bm1 |= v; else L.bm0 |= v; }
PS_HD void bm_clr(BtLane &L, int b) { const unsigned long long v = ~(1ull << (b & 63)); if (b & 64) L.bm1 &= v; else L.bm0 &= v; }
PS_HD int bm_first(const BtLane &L)
{
#ifdef __HIP_DEVICE_COMPILE__
    return L.bm0 ? (__ffsll((unsigned long long)L.bm0) - 1) : (64 + __ffsll((unsigned long long)L.bm1) - 1);
#else
    return L.bm0 ? __builtin_ctzll(L.bm0) : 64 + __builtin_ctzll(L.bm1);
#endif
}
PS_HD void store16(Entry16 *dst, const Entry16 &e)
{
#ifdef __HIP_DEVICE_COMPILE__
    ps_u32x4 v; v.x = e.k; v.y = e.l; v.z = e.a; v.w = e.b;
    *PS_AS_GLOBAL_W(ps_u32x4, dst) = v;        // plain store: non-temporal ones were measured 3 % slower (profiles/r02_kernel_experiments.txt)
#else
    *dst = e;
#endif
}
PS_HD void load16(const Entry16 *src, Entry16 &e)
{
#ifdef __HIP_DEVICE_COMPILE__
    const ps_u32x4 v = *PS_AS_GLOBAL(ps_u32x4, src);
    e.k = v.x; e.l = v.y; e.a = v.z; e.b = v.w;
#else
    e = *src;
#endif
}
PS_HD void store_entry(Entry *dst, const Entry &e)
{
#ifdef __HIP_DEVICE_COMPILE__
    const uint4 *s = reinterpret_cast<const uint4 *>(&e);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    d[0] = s[0]; d[1] = s[1];
#else
    *dst = e;
#endif
}
PS_HD void load_entry(const Entry *src, Entry &e)
{
#ifdef __HIP_DEVICE_COMPILE__
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(&e);
    d[0] = s[0]; d[1] = s[1];
#else
    e = *src;
#endif
}

// wide stack (last tier): push a child entry on its score bucket (LIFO linked list through the entries' next field)
PS_HD void bt_push_wide(const BtHot &a, BtLane &L, BtMem &m, bool want, int i, bwtint k, bwtint l, int n_mm, int n_gapo, int n_gape,
                        int n_ins, int n_del, int state, bool is_diff, int score, int units)
{
    if (!want || units > L.max_units) return;        // unaffordable children are not pushed (no-op with stock costs)
    if (score >= a.n_buckets()) { L.status = RS_BAD_SCORE; return; }
    Entry *pool = reinterpret_cast<Entry *>(m.pool);
    uint32_t idx;
    if (L.free_head != PS_NIL) { idx = L.free_head; L.free_head = pool[idx].next; }
    else if (L.bump < a.pool_cap) idx = L.bump++;
    else { L.status = RS_OVERFLOW_POOL; return; }
    Entry e;
    e.k = k; e.l = l; e.score = (uint16_t)score; e.units = (uint16_t)units;
    e.i = (uint8_t)i; e.last_diff_pos = (uint8_t)(is_diff ? i : 0);
    e.n_mm = (uint8_t)n_mm; e.n_gapo = (uint8_t)n_gapo; e.n_gape = (uint8_t)n_gape;
    e.n_ins = (uint8_t)n_ins; e.n_del = (uint8_t)n_del; e.state = (uint8_t)state;
    e.next = bm_test(L, score) ? m.heads[score] : PS_NIL;
    store_entry(&pool[idx], e);
    m.heads[score] = idx;
    bm_set(L, score);
    ++L.n_stack; ++L.st.pushes;
}

// pop the newest entry of the lowest non-empty score bucket into the lane's current-entry registers
// (wide tier; the narrow tiers' pop is nt_pop in ps_narrow.h)
PS_HD void bt_pop(const BtHot &a, BtLane &L, BtMem &m)
{
    const int b = bm_first(L);
    Entry *pool = reinterpret_cast<Entry *>(m.pool);
    uint32_t h = m.heads[b];
    Entry e;
    load_entry(&pool[h], e);
    if (e.next == PS_NIL) bm_clr(L, b); else m.heads[b] = e.next;
    pool[h].next = L.free_head; L.free_head = h;
    L.k = e.k; L.l = e.l; L.i = e.i; L.score = e.score; L.units = e.units;
    L.n_mm = e.n_mm; L.n_gapo = e.n_gapo; L.n_gape = e.n_gape; L.n_ins = e.n_ins; L.n_del = e.n_del;
    L.state = e.state; L.ldp = e.last_diff_pos;
    (void)a;
    --L.n_stack; ++L.st.pops;
}

PS_HD void bt_finish_read(const BtArgs &a, BtLane &L)
{
    a.n_aln[L.r] = L.n_aln;
    a.status[L.r] = (uint8_t)L.status;
    if (a.read_iters) a.read_iters[PS_RI_WORDS * (size_t)L.r] = L.st.iters - L.iters0;
    L.mode = M_FETCH;
}

// a hit: SA interval [k,l] reached with the current entry's edit counts (upstream bwt_match_gap's hit block)
PS_COLD void bt_hit(const BtArgs &a, BtLane &L, BtMem &m)
{
    const Model &md = a.md;
    if (L.n_aln == 0) {
        L.best_score = L.score;
        int t = L.units + md.u_tight;
        L.max_units = t > L.max_units ? L.max_units : t;     // L.max_units: still the read's own budget before the first hit
    }
    if (L.score == L.best_score) L.best_cnt += (unsigned long long)(L.l - L.k) + 1ull;
    else if (L.best_cnt > (unsigned long long)md.max_top2) { bt_finish_read(a, L); return; }
    bool do_add = true;
    AlnRec *out = a.alns + (size_t)L.r * a.aln_cap;
    if (L.n_gapo) {
        for (int j = 0; j < L.n_aln; ++j)
            if (out[j].k == L.k && out[j].l == L.l) { do_add = false; break; }
    }
    if (!do_add) return;
    // shadow: discount this hit's occurrences from the width bounds left of the last difference
    {
        const uint32_t shadow = a.ix.seq_len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a.ix.seq_len;   // marks stay above every real width
        uint32_t x = (uint32_t)(L.l - L.k + 1), j = 0, prev = 0;
        int lim = L.ldp;
        for (int i = 0; i < lim; ++i) {
            size_t off = (size_t)i * a.n_reads + L.r;
            uint32_t w = a.w[off];
            int bid = m.cw[i] & 0x7f;
            if (w > x) { w -= x; a.w[off] = w; }
            else if (w == x) {
                // positions left of the hit's last difference matched exactly, so their bound was 0 and only ever
                // rises; should a bound drop after all, children skipped on the strength of it (n_phantom) might have
                // been viable: the read is redone by the wide tier, which skips nothing
                if (bid > 1 && !a.wide && L.n_phantom) L.status = RS_OVERFLOW_POOL;
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
    if (L.n_aln >= a.aln_cap) { L.status = RS_OVERFLOW_ALN; bt_finish_read(a, L); return; }
    AlnRec rec;
    rec.k = L.k; rec.l = L.l; rec.score = (uint16_t)L.score; rec.units = (uint16_t)L.units;
    rec.n_mm = (uint8_t)L.n_mm; rec.n_gapo = (uint8_t)L.n_gapo; rec.n_gape = (uint8_t)L.n_gape;
    rec.n_ins = (uint8_t)L.n_ins; rec.n_del = (uint8_t)L.n_del;
    for (int j = 0; j < 7; ++j) rec.pad[j] = 0;
    out[L.n_aln++] = rec;
}

// load one read into the lane's local memory and reset the search state; false if the read is rejected outright
PS_COLD bool bt_fetch(const BtArgs &a, BtLane &L, BtMem &m, int fetch_r)
{
    const Model &md = a.md;
        const int r = a.order ? a.order[fetch_r] : fetch_r;
        const int len = a.lens ? a.lens[r] : a.len;
        L.len = len;
        L.r = r; L.status = RS_OK; L.n_aln = 0; L.iters0 = L.st.iters;
        // load the compact widths and the packed read into local memory (whole words, coalesced across lanes)
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
        const int own_units = (a.lens && a.units_by_len) ? (int)a.units_by_len[len] : md.max_units;     // the read's own budget (md: the longest read's)
        if (nNu > own_units) { bt_finish_read(a, L); return false; }
        L.k = 0; L.l = a.ix.seq_len; L.i = len; L.score = 0; L.units = 0;
        L.n_mm = L.n_gapo = L.n_gape = L.n_ins = L.n_del = 0; L.state = ST_M; L.ldp = 0;
        L.have_cur = true; L.n_stack = 0; L.bm0 = L.bm1 = 0; L.bump = 0; L.free_head = PS_NIL; L.cap = a.pool_cap; L.n_phantom = 0;
        L.best_score = 1 << 29; L.max_units = own_units; L.best_cnt = 0;
        return true;
}

// One iteration of a lane.  fetch_r: the read this lane may take if it is idle (M_FETCH): < 0 = none offered
// now (stay idle), >= n_reads = the input is exhausted (retire), else the read index.
// serve_hit: lanes that reached a hit (M_HIT) record it now; the kernel batches this rare, long path over
// several lanes of a wave instead of running it for one lane at a time.
// (the wide tier; the narrow tiers run nt_iter, ps_narrow.h.)
PS_HD void bt_iter(const BtArgs &a, const BtHot &h, BtLane &L, BtMem &m, int fetch_r, bool serve_hit)
{
    if (L.mode == M_EXIT || L.mode == M_GROW) return;   // retired lane / lane waiting for a larger stack: nothing to do here
    ++L.st.iters;
    if (L.mode == M_HIT) {
        if (!serve_hit) return;
        L.mode = M_POP;
        { BtLane t = L; bt_hit(a, t, m); L = t; }   // may finish the read (mode becomes M_FETCH)
        return;
    }
    if (L.mode == M_FETCH) {
        if (fetch_r < 0) return;
        if (fetch_r >= (int)h.n_reads) { L.mode = M_EXIT; return; }
        { BtLane t = L; const bool ok = bt_fetch(a, t, m, fetch_r); L = t; if (!ok) return; }   // by-value round trip: only the copy is address-taken
        L.mode = M_POP;
    }
    if (L.mode == M_POP) {
        // A popped entry that the bounds reject right away (budget, D(i) bound) costs the lane nothing but the
        // pop: retry within the iteration, a wasted wave iteration is far dearer than a second pop.
#pragma unroll 1
        for (int tries = 0; tries < PS_POP_TRIES; ++tries) {
            int n_virtual = L.n_stack + (L.have_cur ? 1 : 0);
            if (n_virtual == 0 || n_virtual > (int)h.max_entries || L.status != RS_OK) { bt_finish_read(a, L); return; }
            if (L.have_cur) L.have_cur = false;
            else bt_pop(h, L, m);
            if (L.score > L.best_score + h.s_stop()) { bt_finish_read(a, L); return; }
            const int rem = L.max_units - L.units;
            if (rem < 0) continue;
            const int mleft = (int)(((uint32_t)rem * h.inv_c_min) >> 16);     // rem / c_min
            if (L.i > 0 && mleft < (int)(m.cw[L.i - 1] & 0x7f)) continue;
            if (L.i == 0) { L.mode = M_HIT; return; }
            if (mleft == 0 && (L.state == ST_M || h.mode_gape() || L.n_gape == h.max_gape())) L.mode = M_EXACT;
            else L.mode = M_EXPAND;
            break;
        }
    }
    if (L.mode != M_EXACT && L.mode != M_EXPAND) return;
    const int len = L.len, max_len = h.len();      // the read's own length; the launch's longest (layout of the local memory)
    // ---- the memory step, shared by both search modes: Occ(k-1,.) and Occ(l,.) -> the four child intervals ----
    uint32_t ck[4], cl[4];
    occ_pair4(h.blocks, h.primary, L.k, L.l, ck, cl, L.st);
    // child interval of text symbol c: rows [L2[c]+ck[c]+1, L2[c]+cl[c]], non-empty iff ck[c] < cl[c]
    if (L.mode == M_EXACT) {          // no difference left: extend exactly, one base per iteration
        const int c = seq_at(m, L, L.i - 1, len, max_len);
        ++L.st.exact;
        if (c > 3) { L.mode = M_POP; return; }
        const uint32_t ok = sel4(ck, c), ol = sel4(cl, c);
        if (ok >= ol) { L.mode = M_POP; return; }
        const bwtint base = h.L2_dyn(c);
        L.k = base + ok + 1; L.l = base + ol; --L.i;
        if (L.i == 0) L.mode = M_HIT;
        return;
    }
    {
        const int i = L.i - 1;
        ++L.st.nodes;
        const bwtint occ = L.l - L.k + 1;
        const int rem = L.max_units - L.units;
        const int mleft = (int)(((uint32_t)rem * h.inv_c_min) >> 16);
        const int srem = h.seed_units() - L.units;                  // seed budget, in units like the read budget
        const int m_seed = srem <= 0 ? 0 : (int)(((uint32_t)srem * h.inv_c_min) >> 16);
        bool allow_diff = true, allow_M = true;
        const int bnd_same = i > 0 ? (int)(m.cw[i - 1] & 0x7f) : 0;   // D bound a child at position i must still afford
        const int bnd_del = (int)(m.cw[i] & 0x7f);                    // ... and a deletion child (it stays at i+1)
        if (i > 0) {
            int b1 = bnd_same, b0 = bnd_del;
            bool eq = (m.cw[i] & 0x80) != 0;
            if (b1 > mleft - 1) allow_diff = false;
            else if (b1 == mleft - 1 && b0 == mleft - 1 && eq) allow_M = false;
            if (h.use_seed() && len > h.seed_len()) {            // the read's own seed rule
                int ii = i - (len - h.seed_len());
                if (ii > 0) {
                    int s1 = m.csw[ii - 1] & 0x7f, s0 = m.csw[ii] & 0x7f;
                    bool seq_ = (m.csw[ii] & 0x80) != 0;
                    if (s1 > m_seed - 1) allow_diff = false;
                    else if (s1 == m_seed - 1 && s0 == m_seed - 1 && seq_) allow_M = false;
                }
            }
        }
        const int e_mm = L.n_mm, e_go = L.n_gapo, e_ge = L.n_gape, e_ni = L.n_ins, e_nd = L.n_del;
        const int e_sc = L.score, e_un = L.units, e_st = L.state;
        const bwtint ek = L.k, el = L.l;
        const int tmp = e_go + e_ge;
        const int s = seq_at(m, L, i, len, max_len);
        const bool gap_ok = allow_diff && i >= h.indel_end_skip() + tmp && len - i >= h.indel_end_skip() + tmp;
        if (gap_ok) {
            if (e_st == ST_M) {
                if (e_go < h.max_gapo()) {
                    bt_push_wide(h, L, m, true, i, ek, el, e_mm, e_go + 1, e_ge, e_ni + 1, e_nd, ST_I, true, e_sc + h.s_gapo_ins(), e_un + h.u_gapo_ins());
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bt_push_wide(h, L, m, ck[j] < cl[j], i + 1, h.L2(j) + ck[j] + 1, h.L2(j) + cl[j], e_mm, e_go + 1, e_ge, e_ni, e_nd + 1, ST_D, true, e_sc + h.s_gapo_del(), e_un + h.u_gapo_del());
                }
            } else if (e_st == ST_I) {
                if (e_ge < h.max_gape())
                    bt_push_wide(h, L, m, true, i, ek, el, e_mm, e_go, e_ge + 1, e_ni + 1, e_nd, ST_I, true, e_sc + h.s_gape(), e_un + h.u_gape());
            } else {
                if (e_ge < h.max_gape() && ((e_ge + e_go) * h.u_tight() < L.max_units || occ < (bwtint)h.max_del_occ())) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bt_push_wide(h, L, m, ck[j] < cl[j], i + 1, h.L2(j) + ck[j] + 1, h.L2(j) + cl[j], e_mm, e_go, e_ge + 1, e_ni, e_nd + 1, ST_D, true, e_sc + h.s_gape(), e_un + h.u_gape());
                }
            }
        }
        L.mode = M_POP;
        if (allow_diff && allow_M) {
#pragma unroll
            for (int j = 1; j <= 4; ++j) {
                const int c = (s + j) & 3;
                const bool is_mm = (j != 4 || s > 3);
                const uint32_t okc = sel4(ck, c), olc = sel4(cl, c);
                const bwtint base = h.L2_dyn(c), k2 = base + okc + 1, l2 = base + olc;
                const bool ok = okc < olc;
                bt_push_wide(h, L, m, ok && is_mm, i, k2, l2, e_mm + 1, e_go, e_ge, e_ni, e_nd, ST_M, true, e_sc + cost_of(h.s_pk, s, c), e_un + cost_of(h.u_pk, s, c));
                if (ok && !is_mm) { // the match child has the parent's score and is pushed last: it is the next pop
                    L.k = k2; L.l = l2; L.i = i; L.state = ST_M; L.ldp = 0; L.have_cur = true;
                }
            }
        } else if (s < 4) {
            const uint32_t okc = sel4(ck, s), olc = sel4(cl, s);
            const bwtint base = h.L2_dyn(s);
            if (okc < olc) { L.k = base + okc + 1; L.l = base + olc; L.i = i; L.state = ST_M; L.ldp = 0; L.have_cur = true; }
        }
    }
}

// ------------------------------------------------------- SA row -> text ---
// sampled suffix array entry t (row t*sa_intv); entry 0 is the row of the empty suffix and stands for -1
PS_HD bwtint sa_sample(const IndexView &ix, bwtint t)
{
    if (t == 0) return ~(bwtint)0;
    return (bwtint)ix.sa[t] | ((bwtint)((ix.sa_hi[t >> 5] >> (t & 31)) & 1u) << 32);
}
// Walk LF until a sampled row (upstream bwt_sa): one block load per step.
PS_HD bool sa_walk_step(const IndexView &ix, bwtint &row, uint32_t &steps, LaneStats &st)
{
    if ((row & (bwtint)(ix.sa_intv - 1)) == 0) return false;
    ++steps; ++st.lf;
    if (row == ix.primary) { row = 0; return true; }
    int pos = 0;
    const uint32_t b = blk_of(row_to_stored(ix.primary, row), pos);
    Blk x;
    load_blk(ix.blocks, b, x);
    int c = blk_sym(x, pos);
    row = L2_of(ix, c) + blk_count1(x, pos + 1, c);
    return true;
}

// ------------------------------------------- banded global alignment (DP) --
// Restates the banded affine-gap global alignment `bwa samse` runs on gapped
// hits (match 1, mismatch -3, N -1, open 5, extend 1; band w).  H/E rows live
// in fast per-lane memory (LDS in the kernel, stride hs), the 1-byte traceback
// matrix z in global memory (stride zs, lane-interleaved).  Returns n_cigar;
// cigar[j] = len<<4|op (0 M, 1 I, 2 D), at most cap entries.
PS_HD int pac_base(const uint8_t *pac, bwtint p) { return (pac[p >> 2] >> ((~p & 3u) << 1)) & 3; }

template <class QF>
PS_HD int banded_global(int qlen, QF query, int tlen, const uint8_t *pac, bwtint rb, int w,
                        int32_t *H, int32_t *E, int hs, uint8_t *z, size_t zs, uint32_t *cigar, int cap)
{
    const int NEG = -0x40000000, gapo = 5, gape = 1, gapoe = 6;
    int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1, j;
    H[0] = 0; E[0] = NEG;
    for (j = 1; j <= qlen && j <= w; ++j) { H[j * hs] = -(gapo + gape * j); E[j * hs] = NEG; }
    for (; j <= qlen; ++j) H[j * hs] = E[j * hs] = NEG;
    for (int i = 0; i < tlen; ++i) {
        int32_t f = NEG, h1;
        int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
        int t = pac_base(pac, rb + (bwtint)i);
        h1 = beg == 0 ? -(gapo + gape * (i + 1)) : NEG;
        for (j = beg; j < end; ++j) {
            int32_t h = H[j * hs], e = E[j * hs];
            int q = query(j);
            uint8_t d;
            H[j * hs] = h1;
            h += q > 3 ? -1 : (q == t ? 1 : -3);
            d = h >= e ? 0 : 1; h = h >= e ? h : e;
            d = h >= f ? d : 2; h = h >= f ? h : f;
            h1 = h;
            h -= gapoe; e -= gape;
            d |= e > h ? 1 << 2 : 0; e = e > h ? e : h;
            E[j * hs] = e;
            f -= gape;
            d |= f > h ? 2 << 4 : 0; f = f > h ? f : h;
            z[((size_t)i * n_col + (j - beg)) * zs] = d;
        }
        H[end * hs] = h1; E[end * hs] = NEG;
    }
    int n = 0, which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
#define PS_PUSHC(OP, LEN) do { if (n && (cigar[n - 1] & 0xfu) == (uint32_t)(OP)) cigar[n - 1] += (uint32_t)(LEN) << 4; \
        else if (n < cap) cigar[n++] = ((uint32_t)(LEN) << 4) | (uint32_t)(OP); } while (0)
    while (i >= 0 && k >= 0) {
        which = (z[((size_t)i * n_col + (k - (i > w ? i - w : 0))) * zs] >> (which << 1)) & 3;
        if (which == 0) { PS_PUSHC(0, 1); --i; --k; }
        else if (which == 1) { PS_PUSHC(2, 1); --i; }
        else { PS_PUSHC(1, 1); --k; }
    }
    if (i >= 0) PS_PUSHC(2, i + 1);
    if (k >= 0) PS_PUSHC(1, k + 1);
#undef PS_PUSHC
    for (int a = 0; a < n >> 1; ++a) { uint32_t t = cigar[a]; cigar[a] = cigar[n - 1 - a]; cigar[n - 1 - a] = t; }
    return n;
}

}  // namespace ps
