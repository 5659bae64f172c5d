// ps_types.h -- plain data shared by host code and gfx950 kernels.
//
// Hot path replaced: the `bwa parasuite` / `bwa aln` + `bwa samse` child
// processes spawned by /root/reference/src/src/mapping/PARAsuiteMapping.java:63-92
// and BWAMapping.java:51-75.  Data layout is this project's own (the Java side
// only probes that <ref>.bwt exists, PARAsuiteMapping.java:45-46).
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef __HIPCC__
#define PS_HD __host__ __device__ __forceinline__
#define PS_COLD __host__ __device__ inline __attribute__((noinline))   // rare, long paths: real calls keep their
                                                                // scalars out of the hot loop's register budget
#else
#define PS_HD inline
#define PS_COLD inline
#endif

namespace ps {

// Row / text coordinate of the BWT of forward+reverse-complement text.  Rows are 33-bit quantities
// (2*l_pac < 2^33: genomes up to 4.29 Gbp, hg19 has n = 6.27e9 rows): 64-bit in registers, 32 bits + one
// packed high bit wherever they are stored in bulk (narrow stack entries, sampled SA).  Occ counts of one
// symbol stay below 2^32 (checked when the index is built), so block counters and interval sizes are 32-bit.
typedef uint64_t bwtint;
static const bwtint PS_MAX_ROWS = (1ull << 33) - 64;

// ---- FM index: one 64-byte block per 192 BWT symbols --------------------
// cnt[c] = occurrences of c in all earlier blocks; sym[0..5] = low bit plane,
// sym[6..11] = high bit plane of the 192 symbols (symbol p: bit p&31 of word
// p>>5) -- one prefix mask + three popcounts per 32 symbols give all four
// counts.  64 B = half an L2 line; 5.33 bits per base.
static const int PS_BLK_SYMS = 192;
struct OccBlock { uint32_t cnt[4]; uint32_t sym[12]; };

struct IndexView {
    const OccBlock *blocks;   // n_blocks
    const uint32_t *sa;       // low 32 bits of SA[row] for row % sa_intv == 0 (row over T$: n+1 rows); entry 0 stands for -1
    const uint32_t *sa_hi;    // bit 32 of the same samples, 32 per word (stored right behind the low words)
    const uint8_t  *pac;      // forward strand, 2 bit, base p at byte p>>2, bits ((~p)&3)<<1
    bwtint seq_len;           // n = 2*l_pac
    bwtint primary;           // row of the '$' character in the last column
    bwtint l_pac;
    bwtint L2[5];
    uint32_t n_blocks, n_sa;
    int sa_intv;
    const uint32_t *jump;     // child intervals of every string shorter than jump_levels symbols (ps_core.h, "jump table"); derived data, never on disk
    int jump_levels;          // 0: none
};

// ---- cost model for one read length (host fills, kernels read) ----------
// Stock BWA: every edit costs 1 budget unit; scores 3/11/4.  PAR-CLIP profile
// mode: units == score == profile-derived integer cost (our model; oracle/ps_oracle.h).
struct Model {
    uint8_t u_mm[5][4], s_mm[5][4];   // [search-orientation read code][text char]
    uint32_t u_mm_pk[5], s_mm_pk[5];  // the same, one word per read code (byte c = text char): register selects on the device
    int32_t u_gapo_ins, s_gapo_ins, u_gapo_del, s_gapo_del, u_gape, s_gape;
    int32_t s_stop, u_tight, c_min, inv_c_min /* ceil(65536/c_min): x/c_min == x*inv>>16 for the small x used */, max_units, n_buckets;
    int32_t max_gapo, max_gape, mode_gape, indel_end_skip, max_del_occ, max_entries;
    int32_t seed_len, max_seed_diff, max_top2, use_seed, len, profile;
};
static const int PS_MAX_BUCKETS = 128;
static const int PS_MAX_LEN = 250;

// ---- backtracking stack entry (32 B, two 16-B stores) --------------------
struct Entry {
    bwtint k, l;
    uint16_t score, units;
    uint8_t i, last_diff_pos, n_mm, n_gapo, n_gape, n_ins, n_del, state;
    uint32_t next;
};
static const uint32_t PS_NIL = 0xFFFFFFFFu;
enum { ST_M = 0, ST_I = 1, ST_D = 2 };

// one SA-interval hit (what upstream keeps in a .sai record)
struct AlnRec {
    bwtint k, l;
    uint16_t score, units;
    uint8_t n_mm, n_gapo, n_gape, n_ins, n_del, pad[7];
};  // 32 B

enum { RS_OK = 0, RS_OVERFLOW_POOL = 1, RS_OVERFLOW_ALN = 2, RS_BAD_SCORE = 3,
       RS_OVERFLOW_DEEP = 4 };   // the stack outgrew the largest narrow stack there is (65,535 entries: a large slot or the second tier): only the wide tier can hold it

struct KStats {            // per-launch counters (roofline accounting)
    unsigned long long occ_pairs, occ_same_blk, nodes, pushes, pops, lf_steps, iters, exact_steps;
};

static const int PS_RI_WORDS = 20;                     // words per read of the optional per-read profile (BtArgs::read_iters)
// per-batch device state of the backtracking stage (all SoA over reads, or per lane)
struct BtArgs {
    IndexView ix;
    Model md;
    int n_reads, len, n_lanes;                        // len: the longest read of the launch (layout); lens: every read's own
    const int32_t *lens;                              // nullptr: all reads have length len
    const uint8_t *units_by_len;                      // with lens: the difference budget (in units) of a read of every length 0..255; md.max_units is that of the longest
    // reads: 2-bit bases [w][n_reads] (base j of a read: word j>>4, bits 2*(j&15)), N mask [j>>5][n_reads]
    const uint32_t *bases; const uint32_t *nmask; int n_bw, n_mw;
    // from the width kernel, [pos][n_reads]: interval sizes w (updated by hit shadowing), compact
    // width bytes cwb (bid | eq<<7) for the read and cswb for its seed
    uint32_t *w; const uint32_t *cwb; const uint32_t *cswb;   // cwb/cswb: 4 positions per word, [word][n_reads]
    // outputs
    AlnRec *alns; int aln_cap; int32_t *n_aln; uint8_t *status;
    // per-lane scratch
    void *pool; uint32_t pool_cap;                    // per lane: pool_cap entries (16 B narrow / 32 B wide)
    uint32_t *heads;                                  // wide stack only: heads[lane*PS_MAX_BUCKETS + bucket]
    int wide;
    // large stack slots for the few reads that outgrow their private slice (narrow stack only): claimed with a
    // compare-and-swap on the slot's busy word, the private entries are copied over by the whole wave (indices
    // stay valid), released when the read is done
    uint8_t *big_pool; uint32_t big_cap, n_big; uint32_t *big_busy; uint32_t *big_next;
    uint32_t *queue;                                  // next unassigned read (waves take chunks of it)
    const int32_t *order;                             // optional: the read at every queue position (heaviest estimated search first, ps_pipeline.hip run_search); nullptr: position == read
    const uint8_t *est; const uint16_t *est_ab;       // the effort estimate the order was made from (est: the predicted score of the best hit, in budget units); est_ab, optional (profiling): its two scans
    int cap_est;                                      // first tier, profile costs: children that can only matter if the best hit is worse than est are not stored (ps_narrow.h: nt_tail); 1 + what tests subtract from est
    uint32_t *read_iters;                             // optional per-read profile (narrow tiers' counting kernel), PS_RI_WORDS words per read: iterations spent | stack slots used |
                                                      // D bound of the whole read, of the seed <<8, the estimate's two scans <<16, <<24 | best score, final budget <<8, hits <<16 | 16 words: the D bounds
    int hit_min;                                      // lanes with a pending hit a wave collects before it records them
    int fetch_min;                                    // idle lanes a wave waits for before it loads new reads
    KStats *stats;
};

}  // namespace ps
