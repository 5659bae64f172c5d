// ps_kernels.hip -- gfx950 kernels of the mapping hot path.
//
// Replaces the compute inside the `bwa aln|parasuite` and `bwa samse` child
// processes of /root/reference/src/src/mapping/PARAsuiteMapping.java:63-92.
// All four kernels are integer/lookup bound (no MFMA): one read (or one SA
// row) per lane, 64-byte Occ blocks fetched as four 16-byte loads, per-lane
// search state in LDS, loops flattened so that every lane of a wave issues its
// random HBM load in the same iteration.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "ps_core.h"
#include "ps_narrow.h"
#include "ps_kernels.h"

namespace ps {

static const int PS_Q_CHUNK = 64;
#ifndef PS_BT_WAVES
#define PS_BT_WAVES 4      // resident waves per SIMD the narrow search kernel is compiled for (register budget)
#endif

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void flush_stats(KStats *ks, const LaneStats &st)
{
    if (!ks) return;
    unsigned long long v[8] = {st.pairs, st.same, st.nodes, st.pushes, st.pops, st.lf, st.iters, st.exact};
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(ks);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned long long s = wave_sum(v[j]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(dst + j, s);
    }
}

// ---- width stage: D(i) bounds for the read and for its seed ---------------
__global__ void __launch_bounds__(256) k_width(WidthArgs a)
{
    const int stride = gridDim.x * blockDim.x;
    LaneStats st = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += stride) {
        WChain A, B;
        wchain_init(a.ix, A); wchain_init(a.ix, B);
        const int len = a.lens ? a.lens[r] : a.len, seed_len = a.seed_len;
        const bool seeded = a.use_seed && len > seed_len;     // the seed rule is the read's own: a launch may hold reads on either side of the seed length
        uint32_t bw = 0, mw = 0, sbw = 0, smw = 0;  // current base / N-mask words of the two chains
        uint32_t cww = 0, csww = 0;                  // compact width bytes being assembled, 4 positions per word
        for (int i = 0; i < len; ++i) {
            int j = len - 1 - i;
            if (i == 0 || (j & 15) == 15) bw = a.bases[(size_t)(j >> 4) * a.n_reads + r];
            if (i == 0 || (j & 31) == 31) mw = a.nmask[(size_t)(j >> 5) * a.n_reads + r];
            int base = ((mw >> (j & 31)) & 1u) ? 4 : (int)((bw >> (2 * (j & 15))) & 3u);
            uint32_t wv; uint8_t cb;
            if (seeded && i < seed_len) {
                int js = seed_len - 1 - i;
                if (i == 0 || (js & 15) == 15) sbw = a.bases[(size_t)(js >> 4) * a.n_reads + r];
                if (i == 0 || (js & 31) == 31) smw = a.nmask[(size_t)(js >> 5) * a.n_reads + r];
                int sbase = ((smw >> (js & 31)) & 1u) ? 4 : (int)((sbw >> (2 * (js & 15))) & 3u);
                uint32_t swv; uint8_t scb;
                wchain_step(a.ix, B, sbase, swv, scb, i == 0, st);
                csww |= (uint32_t)scb << (8 * (i & 3));
                if ((i & 3) == 3) { a.cswb[(size_t)(i >> 2) * a.n_reads + r] = csww; csww = 0; }
            }
            wchain_step(a.ix, A, base, wv, cb, i == 0, st);
            a.w[(size_t)i * a.n_reads + r] = wv;
            cww |= (uint32_t)cb << (8 * (i & 3));
            if ((i & 3) == 3) { a.cwb[(size_t)(i >> 2) * a.n_reads + r] = cww; cww = 0; }
        }
        a.w[(size_t)len * a.n_reads + r] = 0;
        cww |= (uint32_t)cw_pack(A.bid + 1, false) << (8 * (len & 3));
        a.cwb[(size_t)(len >> 2) * a.n_reads + r] = cww;
        if (seeded) {
            csww |= (uint32_t)cw_pack(B.bid + 1, false) << (8 * (seed_len & 3));
            a.cswb[(size_t)(seed_len >> 2) * a.n_reads + r] = csww;
        } else if (a.use_seed) {                     // a read without a seed in a launch with one: its seed bounds are never looked at, but the lane loads the words
            for (int p = 0; p <= (seed_len >> 2); ++p) a.cswb[(size_t)p * a.n_reads + r] = 0u;
        }
    }
    flush_stats(a.stats, st);
}

// ---- seed / backtracking stage --------------------------------------------
// wave-uniform value pinned in a scalar register (a kernel argument would otherwise be re-read from memory
// wherever the compiler runs short of registers -- inside the loop)
__device__ __forceinline__ uint32_t pin32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long pin64(unsigned long long v)
{
    return (unsigned long long)pin32((uint32_t)v) | ((unsigned long long)pin32((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ void pin_hot(BtHot &h, const BtHot &k)
{
    h.blocks = reinterpret_cast<const OccBlock *>(pin64(reinterpret_cast<unsigned long long>(k.blocks)));
    h.primary = pin64(k.primary);
    h.jump = reinterpret_cast<const uint32_t *>(pin64(reinterpret_cast<unsigned long long>(k.jump))); h.jump_levels = pin32(k.jump_levels);
#pragma unroll
    for (int c = 0; c < 4; ++c) h.L2lo[c] = pin32(k.L2lo[c]);
    h.L2hi = pin32(k.L2hi);
#pragma unroll
    for (int c = 0; c < 5; ++c) { h.s_pk[c] = pin32(k.s_pk[c]); h.u_pk[c] = pin32(k.u_pk[c]); }
    h.p0 = pin32(k.p0); h.p1 = pin32(k.p1); h.p2 = pin32(k.p2); h.p3 = pin32(k.p3);
    h.c_min = pin32(k.c_min);
    h.inv_c_min = pin32(k.inv_c_min); h.max_entries = pin32(k.max_entries); h.pool_cap = pin32(k.pool_cap); h.n_reads = pin32(k.n_reads); h.big_cap = pin32(k.big_cap);
}

// The wide tier (last resort: 32-byte entries, free list, bucket heads in HBM, upstream's 2,000,000-entry limit): reads that
// outgrew both narrow tiers.  Reads are handed out dynamically: search effort differs by orders of magnitude between
// reads, so a lane takes a new read as soon as it is done.  A wave reserves chunks of PS_Q_CHUNK reads from one global
// counter (one atomic per chunk) and deals them to its idle lanes by ballot rank; it loads new reads only when fetch_min
// lanes are idle (or nothing else is running), because the load path is executed by the whole wave.
__global__ void __launch_bounds__(256) k_backtrack_wide(const BtArgs *__restrict__ ap, BtHot hk, int lm_stride)
{
    const BtArgs &a = *ap;      // arguments live in device memory: the cold (non-inlined) paths take their address
    BtHot h; pin_hot(h, hk);
    uint32_t *const queue = reinterpret_cast<uint32_t *>(pin64(reinterpret_cast<unsigned long long>(a.queue)));
    const uint32_t n_reads = h.n_reads, fetch_min = pin32((uint32_t)a.fetch_min), hit_min = pin32((uint32_t)a.hit_min);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane_g = blockIdx.x * blockDim.x + threadIdx.x;
    BtMem m;
    bt_mem_bind(m, smem + (size_t)threadIdx.x * lm_stride, h.len(), h.seed_len());
    m.pool = reinterpret_cast<uint8_t *>(a.pool) + (size_t)lane_g * h.pool_cap * sizeof(Entry);
    m.heads = a.heads + (size_t)lane_g * PS_MAX_BUCKETS;
    BtLane L;
    L.mode = M_FETCH; L.r = 0; L.have_cur = false; L.n_stack = 0; L.status = RS_OK; L.n_aln = 0;
    L.st = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const unsigned long long t_start = wall_clock64();
    int q_next = 0, q_end = 0;
    bool exhausted = false;
    for (;;) {
        const bool want = L.mode == M_FETCH;
        const unsigned long long wmask = __ballot(want), lmask = __ballot(L.mode != M_EXIT), hmask = __ballot(L.mode == M_HIT);
        if (lmask == 0) break;
        const bool stalled = (wmask | hmask) == lmask;          // nobody can advance without being served
        const bool serve_hit = hmask != 0 && (__popcll(hmask) >= (int)hit_min || stalled);
        int fetch_r = -1;
        if (wmask) {
            const int cnt = __popcll(wmask);
            if (cnt >= (int)fetch_min || stalled) {
                const int rank = __popcll(wmask & lane_lt);
                int served = 0;
                while (served < cnt) {
                    if (q_next == q_end) {
                        if (exhausted) break;
                        unsigned int base = 0;
                        if (lane == 0) base = atomicAdd(queue, (unsigned int)PS_Q_CHUNK);
                        base = (unsigned int)__shfl((int)base, 0, 64);
                        if (base >= n_reads) { exhausted = true; break; }
                        q_next = (int)base;
                        q_end = (int)base + PS_Q_CHUNK < (int)n_reads ? (int)base + PS_Q_CHUNK : (int)n_reads;
                    }
                    const int take = q_end - q_next < cnt - served ? q_end - q_next : cnt - served;
                    if (want && rank >= served && rank < served + take) fetch_r = q_next + (rank - served);
                    q_next += take; served += take;
                }
                if (want && fetch_r < 0 && exhausted) fetch_r = (int)n_reads;   // nothing left: this lane retires
            }
        }
        bt_iter(a, h, L, m, fetch_r, serve_hit);
    }
    // how long this wave was busy (160 ns units, summed over waves in the otherwise unused lf field): the mean wave
    // time against the kernel time is the share of the launch spent waiting for the last long reads
    if (lane == 0) L.st.lf = (uint32_t)((wall_clock64() - t_start) >> 4);
    flush_stats(a.stats, L.st);
}


// The narrow tiers (16-byte stack entries; ps_narrow.h): the same read hand-out, the lane state packed, and a stack that
// grows inside the launch: a read that outgrows its private slice moves to one of the large slots (wave-cooperative copy).  STATS: per-lane counters for the roofline accounting and the per-read profile -- the timed kernel has none.
template <bool STATS, bool NB32>
__global__ void __launch_bounds__(256, PS_BT_WAVES) k_backtrack_n(const BtArgs *__restrict__ ap, BtHot hk, int lm_stride)
{
    const BtArgs &a = *ap;      // arguments live in device memory: the cold (non-inlined) paths take their address
    BtHot h; pin_hot(h, hk);
    uint32_t *const queue = reinterpret_cast<uint32_t *>(pin64(reinterpret_cast<unsigned long long>(a.queue)));
    const uint32_t n_reads = h.n_reads, fetch_min = pin32((uint32_t)a.fetch_min), hit_min = pin32((uint32_t)a.hit_min);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane_g = blockIdx.x * blockDim.x + threadIdx.x;
    BtMem m;
    bt_mem_bind(m, smem + (size_t)threadIdx.x * lm_stride, h.len(), h.seed_len());
    uint8_t *const pool_private = reinterpret_cast<uint8_t *>(a.pool) + (size_t)lane_g * h.pool_cap * sizeof(Entry16);
    m.pool = pool_private;
    m.heads = nullptr;
    NLane L;
    nl_init(L);
    LaneStats st;
    ls_init(st);
    NtClock clk;
    for (int j = 0; j < 4; ++j) clk.tm[j] = 0;
    clk.t_last = 0;
#ifdef PS_STAMPS
    clk.t_last = __builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63;
    const unsigned long long t_start = STATS ? wall_clock64() : 0ull;
    uint32_t iters0 = 0;
    int q_next = 0, q_end = 0;
    bool exhausted = false;
    for (;;) {
        const int mode = nl_mode(L.ctl);
        const bool want = mode == M_FETCH;
        if (want && (L.ctl & NL_BIG)) {                        // read done on a large slot: hand the slot back
            const unsigned int slot = (unsigned int)((reinterpret_cast<uint8_t *>(m.pool) - a.big_pool) / ((size_t)h.big_cap * sizeof(Entry16)));
            __threadfence();                                   // this lane's stores to the slot land before the next owner's
            atomicExch(a.big_busy + slot, 0u);
            m.pool = pool_private;                             // a new read starts on the lane's private stack slice
            L.ctl &= ~NL_BIG;
        }
        const unsigned long long wmask = __ballot(want), lmask = __ballot(mode != M_EXIT), hmask = __ballot(mode == M_HIT);
        if (lmask == 0) break;
        const bool stalled = (wmask | hmask) == lmask;          // nobody can advance without being served
        const bool serve_hit = hmask != 0 && (__popcll(hmask) >= (int)hit_min || stalled);
        int fetch_r = -1;
        if (wmask) {
            const int cnt = __popcll(wmask);
            if (cnt >= (int)fetch_min || stalled) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)wmask, 0u));   // idle lanes below this one
                int served = 0;
                while (served < cnt) {
                    if (q_next == q_end) {
                        if (exhausted) break;
                        unsigned int base = 0;
                        if (lane == 0) base = atomicAdd(queue, (unsigned int)PS_Q_CHUNK);
                        base = (unsigned int)__shfl((int)base, 0, 64);
                        if (base >= n_reads) { exhausted = true; break; }
                        q_next = (int)base;
                        q_end = (int)base + PS_Q_CHUNK < (int)n_reads ? (int)base + PS_Q_CHUNK : (int)n_reads;
                    }
                    const int take = q_end - q_next < cnt - served ? q_end - q_next : cnt - served;
                    if (want && rank >= served && rank < served + take) fetch_r = q_next + (rank - served);
                    q_next += take; served += take;
                }
                if (want && fetch_r < 0 && exhausted) fetch_r = (int)n_reads;   // nothing left: this lane retires
            }
        }
        unsigned long long gmask = __ballot(mode == M_GROW);
        while (gmask) {                                        // wave-uniform loop over the lanes that need a larger stack
            const int src = __ffsll((unsigned long long)gmask) - 1;
            gmask &= gmask - 1;
            const unsigned int n_big = a.n_big;
            unsigned int slot = n_big;
            if (lane == src) {                                 // claim a free slot: rotating start, bounded probing
                unsigned int at = atomicAdd(a.big_next, 1u) % n_big;
                for (int tries = 0; tries < 256; ++tries) {
                    if (atomicCAS(a.big_busy + at, 0u, 1u) == 0u) { slot = at; break; }
                    at = at + 1u == n_big ? 0u : at + 1u;
                }
            }
            slot = (unsigned int)__shfl((int)slot, src, 64);
            const unsigned int n_copy = (unsigned int)__shfl((int)nl_bump(L), src, 64);
            const unsigned long long from = (unsigned long long)__shfl((long long)reinterpret_cast<unsigned long long>(m.pool), src, 64);
            if (slot < n_big) {
                const uint4 *sp = reinterpret_cast<const uint4 *>(from);
                uint4 *dp = reinterpret_cast<uint4 *>(a.big_pool + (size_t)slot * h.big_cap * sizeof(Entry16));
                for (unsigned int e = (unsigned int)lane; e < n_copy; e += 64u) dp[e] = sp[e];
                __threadfence();                               // the copies of all lanes are visible before the owner pops from them
                if (lane == src) { m.pool = dp; L.ctl = nl_set_mode(L.ctl | NL_BIG, M_EXPAND); }
            } else if (lane == src) L.ctl = nl_set_mode(nl_set_status(L.ctl, RS_OVERFLOW_POOL), M_POP);
        }
        nt_iter<STATS, NB32>(a, h, L, st, m, fetch_r, serve_hit, &clk);
        if (STATS && a.read_iters) {                           // per-read profile: iterations and stack slots used
            const int now = nl_mode(L.ctl);
            if (mode == M_FETCH && now != M_FETCH && now != M_EXIT) {       // a read was taken: its lower bounds as the width stage left them
                iters0 = st.iters - 1u;
                const int len = nl_len(L), sl = h.seed_len();
                a.read_iters[PS_RI_WORDS * (size_t)L.r + 2] = (uint32_t)(m.cw[len - 1] & 0x7f) | ((h.use_seed() && sl > 0 && sl < len ? (uint32_t)(m.csw[sl - 1] & 0x7f) : 0u) << 8) | (a.est_ab ? (uint32_t)a.est_ab[L.r] << 16 : 0u);
                const uint32_t *cw32 = reinterpret_cast<const uint32_t *>(m.cw);        // words 4..19: the D bounds themselves (bid | eq << 7 per position), 64 positions
                for (int p = 0; p < 16; ++p) a.read_iters[PS_RI_WORDS * (size_t)L.r + 4 + p] = p < lm_ncw(len) ? cw32[p] : 0u;
            }
            if (mode != M_FETCH && mode != M_EXIT && now == M_FETCH) {
                uint32_t *ri = a.read_iters + PS_RI_WORDS * (size_t)L.r;
                ri[0] = st.iters - iters0; ri[1] = nl_bump(L);
                ri[3] = (uint32_t)nl_best_score(L) | ((uint32_t)nl_max_units(L) << 8) | ((uint32_t)nl_n_aln(L.ctl) << 16);
            }
        }
    }
    if (STATS) {
        // how long this wave was busy (160 ns units, summed over waves in the otherwise unused lf field): the mean wave
        // time against the kernel time is the share of the launch spent waiting for the last long reads
        if (lane == 0) st.lf = (uint32_t)((wall_clock64() - t_start) >> 4);
#ifdef PS_STAMPS
        // diagnostic build: the four section times of this wave (64-cycle units) replace nodes / pushes / pops / exact
        st.nodes = lane == 0 ? (uint32_t)(clk.tm[0] >> 6) : 0; st.pushes = lane == 0 ? (uint32_t)(clk.tm[1] >> 6) : 0;
        st.pops = lane == 0 ? (uint32_t)(clk.tm[2] >> 6) : 0; st.exact = lane == 0 ? (uint32_t)(clk.tm[3] >> 6) : 0;
#endif
        flush_stats(a.stats, st);
    } else {
        // the timed kernel keeps two counters: the Occ steps it made and those that needed one block only (ps_narrow.h, nt_occ)
        st.nodes = st.pushes = st.pops = st.iters = st.exact = st.lf = 0;
        flush_stats(a.stats, st);
    }
}

// ---- SA row -> text position ----------------------------------------------
__global__ void __launch_bounds__(256) k_sa2pos(IndexView ix, const bwtint *rows, bwtint *out, int n, KStats *stats)
{
    const int stride = gridDim.x * blockDim.x;
    int item = blockIdx.x * blockDim.x + threadIdx.x;
    LaneStats st = {0, 0, 0, 0, 0, 0, 0, 0};
    bool have = false; bwtint row = 0; uint32_t steps = 0;
    while (true) {
        if (!have) {
            if (item >= n) break;
            row = rows[item]; steps = 0; have = true;
        }
        if (!sa_walk_step(ix, row, steps, st)) {
            out[item] = (bwtint)steps + sa_sample(ix, row / (bwtint)ix.sa_intv);
            item += stride; have = false;
        }
    }
    flush_stats(stats, st);
}

// ---- exhaustive self-check of an index against its text (ps_ctx_index_check) ----
// The rows of the BW matrix form ONE cycle under LF, and walking it spells the text backwards.  Cut at the sampled rows it falls into
// n_sa independent arcs: a thread starts at a sampled row r with p = SA[r], and until it reaches the next sampled row it checks, row by
// row, that the BWT symbol there is T[p - 1] (T = forward strand + reverse complement, read from the packed text, NOT from the BWT) and
// steps row = LF(row), p = p - 1; at the sampled row it arrives at, the stored sample must be the p it has counted down to.  Every row
// lies on exactly one arc, so: symbol mismatches == 0, sample mismatches == 0 and rows visited == n + 1 together say that the last
// column is the BWT of this text, that Occ / L2 (which LF is made of) are consistent with it, and that every SA sample is right.
__device__ __forceinline__ int text_sym(const IndexView &ix, bwtint p) { return p < ix.l_pac ? pac_base(ix.pac, p) : 3 - pac_base(ix.pac, 2 * ix.l_pac - 1 - p); }
__global__ void __launch_bounds__(256) k_index_check(IndexView ix, unsigned long long *out)
{
    unsigned long long rows = 0, bad_sym = 0, bad_sa = 0, longest = 0;
    const bwtint n_sa = ix.n_sa;
    for (bwtint t = (bwtint)blockIdx.x * blockDim.x + threadIdx.x; t < n_sa; t += (bwtint)gridDim.x * blockDim.x) {
        bwtint row = t * (bwtint)ix.sa_intv;
        bwtint p = t == 0 ? ix.seq_len : sa_sample(ix, t);        // row 0 is the empty suffix (stored as -1)
        unsigned long long steps = 0;
        do {
            if (p == 0) {                                           // the whole text: its row holds '$' in the last column
                if (row != ix.primary) ++bad_sym;
                row = 0; p = ix.seq_len;
            } else {
                if (row == ix.primary) { ++bad_sym; break; }
                int pos = 0;
                const uint32_t b = blk_of(row_to_stored(ix.primary, row), pos);
                Blk x;
                load_blk(ix.blocks, b, x);
                const int c = blk_sym(x, pos);
                if (c != text_sym(ix, p - 1)) ++bad_sym;
                row = L2_of(ix, c) + blk_count1(x, pos + 1, c);
                --p;
            }
            ++steps;
        } while ((row & (bwtint)(ix.sa_intv - 1)) != 0 && steps < 100000ull);
        if ((row & (bwtint)(ix.sa_intv - 1)) != 0) ++bad_sa;       // an arc that never closes
        else {
            const bwtint t2 = row / (bwtint)ix.sa_intv;
            const bwtint want = t2 == 0 ? ix.seq_len : sa_sample(ix, t2);
            if (want != p) ++bad_sa;
        }
        rows += steps;
        if (steps > longest) longest = steps;
    }
    rows = wave_sum(rows); bad_sym = wave_sum(bad_sym); bad_sa = wave_sum(bad_sa);
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(longest, o, 64); longest = v > longest ? v : longest; }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out + 0, rows); atomicAdd(out + 1, bad_sym); atomicAdd(out + 2, bad_sa); atomicMax(out + 3, longest); }
}
void launch_index_check(const IndexView &ix, unsigned long long *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_index_check, dim3(256 * 16), dim3(256), 0, s, ix, out);
}

// ---- banded global alignment of gapped hits ---------------------------------
// One hit per lane, 64 lanes per block; H/E rows in LDS ([j][lane]), traceback bytes in global memory.
__global__ void __launch_bounds__(64) k_refine(RefineArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int32_t *H = reinterpret_cast<int32_t *>(smem) + threadIdx.x;
    int32_t *E = H + (size_t)(a.len + 2) * 64;
    uint8_t *z = a.zbuf + (size_t)blockIdx.x * a.z_per_block + threadIdx.x;
    for (int it = blockIdx.x * 64 + threadIdx.x; it < a.n_items; it += gridDim.x * 64) {
        RefineItem q = a.items[it];
        const int qlen = a.lens ? a.lens[q.read] : a.len;
        int tlen = qlen + q.ref_shift;
        int d = tlen - qlen; if (d < 0) d = -d;
        int w = (int)(d * 1.5); if (w < 50) w = 50;
        const int r = q.read, len = qlen, n_reads = a.n_reads;
        const uint32_t *bases = a.bases, *nmask = a.nmask;
        const int strand = q.strand;
        uint32_t cig[PS_MAX_CIGAR];
        int n = banded_global(len,
            [&](int j) { int b = read_base(bases, nmask, n_reads, r, strand ? len - 1 - j : j); return b > 3 ? 4 : (strand ? 3 - b : b); },
            tlen, a.ix.pac, q.rb, w, H, E, 64, z, 64, cig, PS_MAX_CIGAR);
        a.n_cigar[it] = n;
        for (int j = 0; j < PS_MAX_CIGAR; ++j) a.cigar[(size_t)it * PS_MAX_CIGAR + j] = j < n ? cig[j] : 0;
    }
}

// ---- launch wrappers ---------------------------------------------------------
void launch_width(const WidthArgs &a, hipStream_t s)
{
    int blocks = (a.n_reads + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_width, dim3(blocks), dim3(256), 0, s, a);
}
// ---- jump table (ps_core.h): one thread per string of level d, the levels one after the other ----
__global__ void __launch_bounds__(256) k_jump_level(BtHot h, uint32_t *table, bwtint seq_len, int d)
{
    const uint32_t n = 1u << (2 * d);
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) jump_fill_slot(h, table, seq_len, d, s);
}
void launch_jump_build(const IndexView &ix, uint32_t *table, int levels, hipStream_t s)
{
    BtArgs a{};
    a.ix = ix; a.ix.jump = nullptr; a.ix.jump_levels = 0;
    BtHot h;
    (void)bt_hot_make(a, h);                     // only the index fields are read by the builder
    for (int d = 0; d < levels; ++d) {
        const uint32_t n = 1u << (2 * d);
        const int blocks = (int)std::min<uint32_t>((n + 255u) / 256u, 256u * 16u);
        hipLaunchKernelGGL(k_jump_level, dim3(blocks), dim3(256), 0, s, h, table, ix.seq_len, d);
    }
}
bool launch_backtrack(const BtArgs &a_in, const BtArgs *d_args, BtArgs *h_stage, int n_blocks, int lm_stride, hipStream_t s, bool stats)
{
    // the patched copy lives in the caller's page-locked staging buffer: it outlives the asynchronous upload whatever the runtime does
    // with pageable sources (the caller synchronises the stream before it touches the buffer again)
    BtArgs &a = *h_stage;
    a = a_in;
    // the counting kernel counts the blocks of the plain algorithm (SURVEY.md 8d's figure), the wide tier keeps whole intervals
    // in its entries: both search without the jump table.  PS_JUMP=0: the timed kernel too (A/B measurements)
    const char *ej = std::getenv("PS_JUMP");
    const bool stats_plain = stats && !std::getenv("PS_STATS_JUMP");     // PS_STATS_JUMP=1: diagnostic runs of the counting / stamps build WITH the table
    if (stats_plain || a.wide || (ej && std::atoi(ej) == 0)) { a.ix.jump = nullptr; a.ix.jump_levels = 0; }
    BtHot h;
    if (!bt_hot_make(a, h)) return false;       // a model field outside its packed range
    (void)hipMemcpyAsync(const_cast<BtArgs *>(d_args), &a, sizeof(BtArgs), hipMemcpyHostToDevice, s);
    const size_t lds = (size_t)256 * lm_stride;
    const bool nb32 = a.md.n_buckets <= 32;        // one-word bucket bitmap (ps_narrow.h)
    typedef void (*KernelN)(const BtArgs *, BtHot, int);
    const KernelN kn = stats ? (nb32 ? k_backtrack_n<true, true> : k_backtrack_n<true, false>) : (nb32 ? k_backtrack_n<false, true> : k_backtrack_n<false, false>);
    const void *fn = a.wide ? reinterpret_cast<const void *>(k_backtrack_wide) : reinterpret_cast<const void *>(kn);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (a.wide) hipLaunchKernelGGL(k_backtrack_wide, dim3(n_blocks), dim3(256), lds, s, d_args, h, lm_stride);
    else hipLaunchKernelGGL(kn, dim3(n_blocks), dim3(256), lds, s, d_args, h, lm_stride);
    return true;
}
void launch_sa2pos(const IndexView &ix, const bwtint *rows, bwtint *out, int n, KStats *stats, hipStream_t s)
{
    int blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_sa2pos, dim3(blocks), dim3(256), 0, s, ix, rows, out, n, stats);
}
void launch_refine(const RefineArgs &a, int n_blocks, hipStream_t s)
{
    size_t lds = (size_t)(a.len + 2) * 64 * 4 * 2;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_refine), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_refine, dim3(n_blocks), dim3(64), lds, s, a);
}

}  // namespace ps
