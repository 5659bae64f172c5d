// ps_kernels.h -- launch interface of the gfx950 kernels (ps_kernels.hip, ps_index.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "ps_types.h"

namespace ps {

static const int PS_MAX_CIGAR = 16;

struct WidthArgs {
    IndexView ix;
    int n_reads, len, seed_len, use_seed;
    const int32_t *lens;     // per-read lengths (nullptr: all len)
    const uint32_t *bases; const uint32_t *nmask;
    uint32_t *w; uint32_t *cwb; uint32_t *cswb;   // compact width bytes, 4 positions per word, [word][n_reads]
    KStats *stats;
};

// Effort estimate (scheduling only, never a result): a read's search effort follows the score of its best hit, which is unknown
// before the search; two greedy scans (from either end of the read) give a cheap estimate of it -- k_effort, ps_kernels.hip
struct EffortArgs {
    IndexView ix;
    int n_reads, len;
    const int32_t *lens;
    const uint32_t *bases; const uint32_t *nmask;
    uint32_t s_pk[5];        // substitution costs as the search uses them (Model::s_mm_pk)
    int c_restart;           // charged where a scan has to start a new piece
    uint32_t w_pin;          // a substitution is believed where the interval is at most this wide (the scan has pinned its locus down)
    uint8_t *est;            // out: estimated score of the best hit, clipped to 255
    uint16_t *est_ab;        // optional (profiling): the two scans' totals (7 bits each) and whether each had to start over (bit 7)
};

// expected search nodes of a read from its estimated final budget and its D(i) bounds (ps_effort.hip); out: the sort key of the hand-out order
struct EffortModelArgs {
    int n_reads, len;
    const int32_t *lens; const uint8_t *units_by_len;
    const uint32_t *bases; const uint32_t *nmask; const uint32_t *cwb;
    const uint8_t *est;
    uint32_t s_pk[5]; uint32_t inv_c_min;
    int max_units, u_tight, seed_units, use_seed, seed_len;
    int max_gapo, indel_end_skip, u_gapo_ins, u_gapo_del;   // gap openings: one insertion child, four deletion children where the search allows an indel
    int depth;               // levels modelled (beyond ~19 symbols a random string no longer occurs in a genome of this size)
    float rows;              // BWT rows: a string of d symbols has min(1, rows / 4^d) expected occurrences
    int log_scale;           // key = 255 - log2(expected nodes) * log_scale
    uint8_t *key; float *pred;      // pred: optional (profiling)
};

struct RefineItem { int32_t read; bwtint rb; int32_t ref_shift; int32_t strand; };
struct RefineArgs {
    IndexView ix;
    int n_items, len, n_reads;             // len: the longest read (sizes the H/E rows)
    const int32_t *lens;                   // per-read lengths (nullptr: all len)
    const uint32_t *bases; const uint32_t *nmask;
    const RefineItem *items;
    uint32_t *cigar; int32_t *n_cigar;     // [n_items][PS_MAX_CIGAR]
    uint8_t *zbuf; size_t z_per_block;     // traceback scratch, 64 lanes interleaved per block
};

void launch_width(const WidthArgs &a, hipStream_t s);
void launch_effort(const EffortArgs &a, hipStream_t s);
void launch_effort_model(const EffortModelArgs &a, hipStream_t s);
// false: model outside the packed ranges.  stats: the narrow tiers' kernel with per-lane counters (KStats, read_iters); the timed kernel carries none
// fills the jump table of an index (ps_core.h): `levels` levels, jump_words(levels) words at `table`
void launch_jump_build(const IndexView &ix, uint32_t *table, int levels, hipStream_t s);
bool launch_backtrack(const BtArgs &a, const BtArgs *d_args /* device copy, filled here */, BtArgs *h_stage /* page-locked, one per stream: the source of that upload */,
                      int n_blocks, int lm_stride, hipStream_t s, bool stats);
void launch_index_check(const IndexView &ix, unsigned long long *out /* device: rows visited, symbol mismatches, sample mismatches, longest arc */, hipStream_t s);
void launch_sa2pos(const IndexView &ix, const bwtint *rows, bwtint *out, int n, KStats *stats, hipStream_t s);
void launch_refine(const RefineArgs &a, int n_blocks, hipStream_t s);

}  // namespace ps
