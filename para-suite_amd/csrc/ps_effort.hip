// ps_effort.hip -- in which order a search launch hands its reads out (gfx950 kernels; scheduling only: no result depends on it).
//
// The search of /root/reference/src/src/mapping/PARAsuiteMapping.java:63-77 (ps_narrow.h) is one read per lane with persistent
// waves, and search effort per read is heavy-tailed: 10 M simulated PAR-CLIP reads need 4,150 iterations on average, 1,220 at
// the median, 24,500 at the 99th percentile and 126,000 at most -- 0.8 s of a 1.26 s launch for that one lane.  Handed out in
// input (or leading-base) order the launch ends with ~0.2 s of emptying machine; handed out longest first (an oracle order by
// the TRUE iteration counts) the same kernel takes 14 % less, 9-12 % with coarse classes that keep the leading-base locality
// (profiles/r03_order_probe.txt).  The true count is what the search computes, but it is predictable:
//
//  1. k_effort: a read's effort follows the budget its search ends with, i.e. the score of its best hit (mean iterations 238 /
//     475 / 1,126 / 3,574 / 14,919 at final budgets 8 / 11 / 14 / 17 / 24).  Two greedy scans guess that score: extend the read
//     exactly through the index; where the interval empties and the scan has pinned its locus down (an interval of at most
//     w_pin rows), take the cheapest substitution that continues it; else start a new piece and charge an average mismatch.  A
//     difference inside the first ~16 bases of a piece is not seen where it is (the interval still holds random matches) and
//     several of them collapse into one restart, so one scan runs from either end of the read (the index holds both strands):
//     a scan that never had to start over is exact; if both had to, the larger total counts.  Three substitutions within a few
//     bases of each other are an indel, a wrong locus or three real differences: they stay charged and the scan goes on with a
//     new piece (the read may be shifted against the text from there on and would pay a substitution per base).  Measured on
//     10 M simulated PAR-CLIP reads: the final budget is hit exactly for 92 % of the reads, over-estimated for 7 %
//     (profiles/r03_order_probe.txt).
//  2. k_effort_model: within one budget the effort still varies 1:10 with the lower bounds D(i) the width stage computed (they
//     prune the search: a read whose differences sit where the bounds cannot see them is searched almost exhaustively).  The
//     number of nodes the search expands is estimated by running its own rules on EXPECTED counts: W[u] = expected live partial
//     alignments with u units spent after d bases, children by the cost table, pruned by exactly the tests of ps_narrow.h
//     (budget, D(i) bound, seed budget, where an indel may open), random continuations weighted by min(1, rows / 4^d).  Against
//     the true iteration counts: r = 0.99 on the log scale with the true budget, 0.93 inside the heaviest budget class, and the
//     200 heaviest of 400,000 reads all land in the first 0.2 % of the order (profiles/r03_order_probe.txt).
//  3. run_search (ps_pipeline.hip) sorts the reads by the quantised log of that number, heaviest first, stable (the given
//     leading-base order inside a class), and the search kernel takes queue position -> read from the result (BtArgs::order).
#include <hip/hip_runtime.h>
#include "ps_core.h"
#include "ps_kernels.h"

namespace ps {

struct EChain {
    bwtint k, l;
    uint32_t cost_lo, cost_hi;     // charges at read positions below / from the middle of the read
    int piece;                     // bases in the current piece
    int n_rs;                      // pieces started over (restarts, suspected indels): what such a scan says about the bases just behind a restart is a guess
    int cl_n, cl_last;             // substitutions of the current cluster, position of the last one
    uint32_t cl_lo, cl_hi;         // the charges before the cluster began
};
__device__ __forceinline__ void echain_restart(const EffortArgs &a, EChain &c, bool lo_half)
{
    c.k = 0; c.l = a.ix.seq_len; c.piece = 0; c.cl_n = 0; ++c.n_rs;
    if (lo_half) c.cost_lo += (uint32_t)a.c_restart; else c.cost_hi += (uint32_t)a.c_restart;
}
// sym: the symbol the pattern grows by (0..3, 4 = N); cw: cost of finding text symbol t there instead (byte t); pos: read position of the base
__device__ __forceinline__ void echain_step(const EffortArgs &a, EChain &c, int sym, uint32_t cw, int pos, bool lo_half, LaneStats &st)
{
    uint32_t ck[4], cl[4];
    occ_pair4(a.ix.blocks, a.ix.primary, c.k, c.l, ck, cl, st);
    if (sym < 4) {
        const uint32_t ok = sel4(ck, sym), ol = sel4(cl, sym);
        if (ok < ol) { const bwtint b = L2_of(a.ix, sym); c.k = b + ok + 1; c.l = b + ol; ++c.piece; return; }
    }
    int best = -1; uint32_t best_cost = 0xffu;
    if (c.l - c.k < (bwtint)a.w_pin) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t ct = (cw >> (8 * t)) & 0xffu;
            if (t != sym && ck[t] < cl[t] && ct < best_cost) { best = t; best_cost = ct; }
        }
    }
    if (best < 0) { echain_restart(a, c, lo_half); return; }
    const int d = pos > c.cl_last ? pos - c.cl_last : c.cl_last - pos;
    if (c.cl_n > 0 && d <= 4) ++c.cl_n; else { c.cl_n = 1; c.cl_lo = c.cost_lo; c.cl_hi = c.cost_hi; }
    c.cl_last = pos;
    if (lo_half) c.cost_lo += best_cost; else c.cost_hi += best_cost;
    if (c.cl_n >= 3) {
        // the third substitution within a few bases: an indel, a wrong locus, or really three differences in a row -- the read is
        // (or may be) shifted against the text from here on and would pay a substitution per base.  The three stay charged (three real
        // differences cost exactly that; an indel costs less, but a read that runs early costs nothing, and the 20 heaviest reads of
        // the bench batch were of this kind and under-charged by a flat gap cost), the scan goes on with a new piece
        c.k = 0; c.l = a.ix.seq_len; c.piece = 0; c.cl_n = 0; ++c.n_rs;
        return;
    }
    const bwtint b = L2_of(a.ix, best);
    c.k = b + sel4(ck, best) + 1; c.l = b + sel4(cl, best); ++c.piece;
}

__global__ void __launch_bounds__(256) k_effort(EffortArgs a)
{
    const int stride = gridDim.x * blockDim.x;
    LaneStats st = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += stride) {
        const int len = a.lens ? a.lens[r] : a.len, half = len >> 1;
        EChain A = {0, a.ix.seq_len, 0, 0, 0, 0, 0, 0, 0, 0}, B = A;
        uint32_t bwA = 0, mwA = 0, bwB = 0, mwB = 0;
        for (int i = 0; i < len; ++i) {
            // chain A: the read itself, grown leftwards from its last base; chain B: its reverse complement, i.e. the read from its first base
            const int ja = len - 1 - i, jb = i;
            if (i == 0 || (ja & 15) == 15) bwA = a.bases[(size_t)(ja >> 4) * a.n_reads + r];
            if (i == 0 || (ja & 31) == 31) mwA = a.nmask[(size_t)(ja >> 5) * a.n_reads + r];
            if ((jb & 15) == 0) bwB = a.bases[(size_t)(jb >> 4) * a.n_reads + r];
            if ((jb & 31) == 0) mwB = a.nmask[(size_t)(jb >> 5) * a.n_reads + r];
            const int ba = ((mwA >> (ja & 31)) & 1u) ? 4 : (int)((bwA >> (2 * (ja & 15))) & 3u);
            const int bb = ((mwB >> (jb & 31)) & 1u) ? 4 : (int)((bwB >> (2 * (jb & 15))) & 3u);
            // costs: the search consumes the reverse-complemented read (code s = 3 - base) against text symbol t: s_pk[s] byte t.
            // Chain B is exactly that.  Chain A matches the other strand: base b against text t is code 3 - b against text 3 - t.
            const int sb = bb > 3 ? 4 : 3 - bb;
            const uint32_t cwB = cost_word(a.s_pk, sb);
            const uint32_t cwA = __builtin_bswap32(cost_word(a.s_pk, ba > 3 ? 4 : 3 - ba));      // byte t of cwA = byte 3 - t of the search's word
            echain_step(a, A, ba, cwA, ja, ja < half, st);
            echain_step(a, B, sb, cwB, jb, jb < half, st);
        }
        // a scan that never lost its locus has priced every difference where it is: its total stands.  One that started over may
        // have folded several differences of its blind zone into one charge (too low) or paid an average mismatch for a cheap
        // conversion there (too high): if both did, the larger total counts -- running a read too early costs nothing, running a
        // heavy one late costs the launch its tail
        const uint32_t ta = A.cost_lo + A.cost_hi, tb = B.cost_lo + B.cost_hi;
        const uint32_t e = A.n_rs == 0 ? (B.n_rs == 0 && tb < ta ? tb : ta) : (B.n_rs == 0 ? tb : (ta > tb ? ta : tb));
        a.est[r] = (uint8_t)(e > 255u ? 255u : e);
        if (a.est_ab) a.est_ab[r] = (uint16_t)((ta > 127u ? 127u : ta) | (A.n_rs ? 0x80u : 0u) | ((tb > 127u ? 127u : tb) << 8) | (B.n_rs ? 0x8000u : 0u));     // profiling: totals, bit 7: the scan started over
    }
}

// expected number of nodes the search expands (see the head of this file); one read per lane, W[] in local memory
__global__ void __launch_bounds__(256) k_effort_model(EffortModelArgs a)
{
    extern __shared__ float smem_f[];
    const int NU = a.max_units + 2;
    float *Wa = smem_f + (size_t)threadIdx.x * 2 * NU, *Wb = Wa + NU;
    const int stride = gridDim.x * blockDim.x;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += stride) {
        const int len = a.lens ? a.lens[r] : a.len;
        const int own = (a.lens && a.units_by_len) ? (int)a.units_by_len[len] : a.max_units;
        int B = (int)a.est[r] + a.u_tight;
        if (B > own) B = own;
        for (int u = 0; u <= a.max_units; ++u) Wa[u] = 0.f;
        Wa[0] = 1.f;
        float tot = 0.f, phi_rows = a.rows;
        const int depth = len < a.depth ? len : a.depth;
        float *W = Wa, *Wn = Wb;
        for (int d = 0; d < depth; ++d) {
            const int i = len - 1 - d;                                   // the search's position: it consumes read base d against seq[i]
            const uint32_t bw = a.bases[(size_t)(d >> 4) * a.n_reads + r], mw = a.nmask[(size_t)(d >> 5) * a.n_reads + r];
            const int base = ((mw >> (d & 31)) & 1u) ? 4 : (int)((bw >> (2 * (d & 15))) & 3u);
            const int s = base > 3 ? 4 : 3 - base;
            const uint32_t cw = cost_word(a.s_pk, s);
            const uint32_t cwd_i = a.cwb[(size_t)(i >> 2) * a.n_reads + r];
            const int D_i = (int)((cwd_i >> (8 * (i & 3))) & 0x7fu);
            int D_im1 = 0;
            if (i > 0) { const uint32_t w2 = a.cwb[(size_t)((i - 1) >> 2) * a.n_reads + r]; D_im1 = (int)((w2 >> (8 * ((i - 1) & 3))) & 0x7fu); }
            const bool seed_chk = a.use_seed && len > a.seed_len && i > 0 && (i - (len - a.seed_len)) > 0;
            const bool gap_here = a.max_gapo > 0 && i >= a.indel_end_skip && len - i >= a.indel_end_skip;
            phi_rows *= 0.25f;                                           // rows / 4^(d+1): expected random continuations of a string of d+1 symbols
            const float phi = phi_rows < 1.f ? phi_rows : 1.f;
            for (int u = 0; u <= B; ++u) Wn[u] = 0.f;
            for (int u = 0; u <= B; ++u) {
                const float w = W[u];
                if (w == 0.f) continue;
                const int rem = B - u;
                const int m = (int)(((uint32_t)rem * a.inv_c_min) >> 16);
                if (m < D_i) continue;                                   // the pop's own test: dropped
                tot += w;
                const float wc = w * phi;
                if (s < 4) Wn[u] += wc;                                  // the match child
                bool allow = i > 0 ? m >= D_im1 + 1 : true;
                if (seed_chk) { const int srem = a.seed_units - u; allow = allow && srem > 0 && (int)(((uint32_t)srem * a.inv_c_min) >> 16) >= 1; }
                if (!allow) continue;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t == s) continue;
                    const int c = (int)((cw >> (8 * t)) & 0xffu);
                    if (u + c <= B) Wn[u + c] += wc;
                }
                if (gap_here) {                                          // gap openings (extensions and the gap states are not modelled)
                    if (u + a.u_gapo_del <= B) Wn[u + a.u_gapo_del] += 4.f * wc;
                    if (u + a.u_gapo_ins <= B) Wn[u + a.u_gapo_ins] += wc;
                }
            }
            float *x = W; W = Wn; Wn = x;
        }
        float left = 0.f;
        for (int u = 0; u <= B; ++u) left += W[u];
        tot += left * (float)(len - depth);                               // what is still alive walks the rest of the read
        const float lg = __log2f(tot + 1.f) * (float)a.log_scale;
        int q = (int)lg; if (q > 255) q = 255; if (q < 0) q = 0;
        a.key[r] = (uint8_t)(255 - q);                                    // ascending sort = heaviest first
        if (a.pred) a.pred[r] = tot;
    }
}

void launch_effort(const EffortArgs &a, hipStream_t s)
{
    int blocks = (a.n_reads + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_effort, dim3(blocks), dim3(256), 0, s, a);
}
void launch_effort_model(const EffortModelArgs &a, hipStream_t s)
{
    int blocks = (a.n_reads + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const size_t lds = (size_t)256 * 2 * (a.max_units + 2) * sizeof(float);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_effort_model), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_effort_model, dim3(blocks), dim3(256), lds, s, a);
}

}  // namespace ps
