// ps_index.hip -- `bwa index` replacement: FASTA -> 2-bit pac, BWT of
// forward+reverse-complement text, 64-byte Occ blocks, SA sampled every 32 rows.
//
// Call site replaced: /root/reference/src/src/mapping/PARAsuiteMapping.java:45-55
// (index if <ref>.bwt is missing).  The suffix array is built ON THE GPU by
// prefix doubling over hipcub radix sorts (keys and ranks stay in HBM; a 1 Gbp
// genome needs ~70 GB of the 288 GB); there is no host suffix sorter.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <atomic>
#include <cctype>
#include <functional>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <mutex>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include "ps_host.h"
#include "ps_core.h"

namespace ps {

// ---------------------------------------------------------------- FASTA -----
static inline int nt4(int c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                 case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

// Non-ACGT characters become pseudo-random bases drawn from lrand48() seeded with 11 and are
// recorded as holes (maximal runs of one identical character), as `bwa index` does.
// A 3 GB FASTA is parsed on several threads: the bodies of the records are cut into pieces of whole lines,
// a first pass counts bases and ambiguous characters per piece, prefix sums give every piece its place in
// the pac, in its contig and in the lrand48 stream (jump-ahead), the second pass packs.
void load_fasta(const char *path, RefSeq &ref)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) throw Error(std::string("cannot open reference ") + path);
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)sz + 1);
    if (sz && std::fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); throw Error(std::string("short read on ") + path); }
    std::fclose(f);
    const size_t n = (size_t)sz;
    const char *b = buf.data();
    ref = RefSeq();
    // ---- records: header fields and body spans
    struct Span { size_t lo, hi; int contig; int64_t n_seq = 0, n_amb = 0, pos0 = 0, amb0 = 0; std::vector<Hole> holes; };
    std::vector<Span> spans;
    size_t PIECE = (size_t)8 << 20;
    if (const char *e = std::getenv("PS_FASTA_PIECE")) PIECE = (size_t)std::max(1, std::atoi(e));     // tests: force many pieces
    size_t i = 0;
    while (i < n) {
        while (i < n && b[i] != '>') ++i;
        if (i >= n) break;
        size_t s = ++i;
        while (i < n && !std::isspace((unsigned char)b[i])) ++i;
        Contig c;
        c.name.assign(b + s, i - s);
        size_t e = i; while (e < n && b[e] != '\n') ++e;
        size_t cs = i; while (cs < e && std::isspace((unsigned char)b[cs])) ++cs;
        size_t ce = e; while (ce > cs && std::isspace((unsigned char)b[ce - 1])) --ce;
        c.anno = ce > cs ? std::string(b + cs, ce - cs) : std::string("(null)");
        c.offset = 0; c.len = 0; c.n_ambs = 0;
        i = e;
        const char *nx = i < n ? (const char *)std::memchr(b + i, '>', n - i) : nullptr;
        const size_t body_end = nx ? (size_t)(nx - b) : n;
        const int ci = (int)ref.contigs.size();
        ref.contigs.push_back(c);
        for (size_t lo = i; lo < body_end;) {                      // pieces end at a line end
            size_t hi = std::min(body_end, lo + PIECE);
            if (hi < body_end) { const char *nl = (const char *)std::memchr(b + hi, '\n', body_end - hi); hi = nl ? (size_t)(nl - b) + 1 : body_end; }
            Span sp; sp.lo = lo; sp.hi = hi; sp.contig = ci;
            spans.push_back(std::move(sp));
            lo = hi;
        }
        i = body_end;
    }
    if (ref.contigs.empty()) throw Error(std::string("no sequences in ") + path);
    int threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    auto run = [&](const std::function<void(Span &)> &fn) {
        std::atomic<size_t> next(0);
        std::vector<std::thread> th;
        auto work = [&]() { for (size_t k; (k = next.fetch_add(1)) < spans.size();) fn(spans[k]); };
        for (int t = 1; t < threads; ++t) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    };
    // ---- pass 1: counts
    run([&](Span &sp) {
        int64_t ns = 0, na = 0;
        for (size_t j = sp.lo; j < sp.hi; ++j) {
            const int ch = (unsigned char)b[j];
            if (!std::isgraph(ch)) continue;
            ++ns; na += nt4(ch) >= 4;
        }
        sp.n_seq = ns; sp.n_amb = na;
    });
    int64_t l_pac = 0, amb = 0;
    for (size_t k = 0; k < spans.size(); ++k) {
        Span &sp = spans[k];
        Contig &c = ref.contigs[(size_t)sp.contig];
        if (k == 0 || spans[k - 1].contig != sp.contig) c.offset = l_pac;
        // contigs without a body keep the running offset
        sp.pos0 = l_pac; sp.amb0 = amb;
        l_pac += sp.n_seq; amb += sp.n_amb;
        if (c.len + sp.n_seq > 0x7fffffffLL) throw Error("a reference sequence is longer than 2^31-1 bases: " + c.name);
        c.len += (int32_t)sp.n_seq;
    }
    { int64_t off = 0; for (Contig &c : ref.contigs) { c.offset = off; off += c.len; } }
    ref.l_pac = l_pac;
    ref.pac.assign((size_t)l_pac / 4 + 2, 0);
    // ---- pass 2: pack (the first and last byte of a piece may be shared with its neighbours)
    run([&](Span &sp) {
        Rng48 rng(11);
        rng.jump((uint64_t)sp.amb0);
        int64_t p = sp.pos0;
        const int64_t first_byte = sp.pos0 >> 2, last_byte = (sp.pos0 + sp.n_seq - 1) >> 2;
        int lasts = 0; bool open_hole = false;
        for (size_t j = sp.lo; j < sp.hi; ++j) {
            const int ch = (unsigned char)b[j];
            if (!std::isgraph(ch)) continue;
            int code = nt4(ch);
            if (code >= 4) {
                if (open_hole && lasts == ch) ++sp.holes.back().len;
                else { sp.holes.push_back(Hole{p, 1, (char)ch}); open_hole = true; }
                code = (int)(rng.lrand() & 3);
            }
            lasts = ch;
            const uint8_t bits = (uint8_t)(code << ((~p & 3) << 1));
            const int64_t byte = p >> 2;
            if (byte == first_byte || byte == last_byte) __atomic_fetch_or(&ref.pac[(size_t)byte], bits, __ATOMIC_RELAXED);
            else ref.pac[(size_t)byte] |= bits;
            ++p;
        }
    });
    // ---- holes in order; a run of one character that continues across a piece boundary is one hole
    for (size_t k = 0; k < spans.size(); ++k) {
        Span &sp = spans[k];
        for (Hole &h : sp.holes) {
            const bool same_contig = k > 0 && spans[k - 1].contig == sp.contig;
            if (!ref.holes.empty() && same_contig && &h == &sp.holes.front() && h.offset == sp.pos0 &&
                ref.holes.back().offset + ref.holes.back().len == h.offset && ref.holes.back().amb == h.amb) {
                // only if the previous piece really ended with this character (no other base in between): offsets adjacent
                ref.holes.back().len += h.len;
            } else { ref.holes.push_back(h); ++ref.contigs[(size_t)sp.contig].n_ambs; }
        }
    }
    ref.pac.resize((size_t)ref.l_pac / 4 + 1);
}

int RefSeq::pos2rid(int64_t pos_f) const
{
    if (pos_f >= l_pac) return -1;
    int left = 0, mid = 0, right = (int)contigs.size();
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= contigs[mid].offset) {
            if (mid == (int)contigs.size() - 1) break;
            if (pos_f < contigs[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
// number of ambiguous reference bases under [pos_f, pos_f+len): the first overlapping hole found by bisection
int RefSeq::cnt_ambi(int64_t pos_f, int len, int *ref_id) const
{
    if (ref_id) *ref_id = pos2rid(pos_f);
    int left = 0, right = (int)holes.size(), nn = 0;
    while (left < right) {
        int mid = (left + right) >> 1;
        const Hole &h = holes[mid];
        if (pos_f >= h.offset + h.len) left = mid + 1;
        else if (pos_f + len <= h.offset) right = mid;
        else {
            if (pos_f >= h.offset) nn += h.offset + h.len < pos_f + len ? (int)(h.offset + h.len - pos_f) : len;
            else nn += h.offset + h.len < pos_f + len ? h.len : len - (int)(h.offset - pos_f);
            break;
        }
    }
    return nn;
}

void Rng48::jump(uint64_t t)
{
    // compose the affine map x -> a*x + c with itself t times (mod 2^48)
    const uint64_t M = 0xFFFFFFFFFFFFULL;
    uint64_t a = 0x5DEECE66DULL, c = 0xBULL, ra = 1, rc = 0;
    while (t) {
        if (t & 1) { ra = (ra * a) & M; rc = (rc * a + c) & M; }
        c = (c * a + c) & M; a = (a * a) & M;
        t >>= 1;
    }
    x = (ra * x + rc) & M;
}

// ------------------------------------------------------ GPU suffix sorting --
// Suffix array of T$ (T = forward + reverse-complement text, n = 2*l_pac symbols, up to 2^33 rows) by prefix
// doubling that only ever re-sorts what is still tied (Larsson-Sadakane style), built from hipcub radix sorts:
//   round 0   the suffixes are split by their first symbol (four chunks, each < 2^32 suffixes); a chunk is
//             sorted by the 27 symbols that follow (one 63-bit base-5 key), so all suffixes are ordered by
//             their first 28 symbols and get the rank of their group's first row
//   round j   only suffixes in groups of two or more remain; they are ordered inside their group by the rank
//             of the suffix h symbols further on (two stable sorts: by that rank, then by group), groups split,
//             singletons drop out, h doubles
// SA and inverse SA (8 bytes per row each) stay in HBM: 6.27e9 rows (hg19) need ~100 GB for them, ~50 GB for
// the chunk being sorted and whatever is still tied -- well inside one MI355X's 288 GB; there is no host sorter.
typedef unsigned long long u64;
static const int GRID = 256 * 8, BLK = 256;

__global__ void k_expand_text(const uint8_t *pac, uint8_t *T, u64 l_pac)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < l_pac; i += (u64)gridDim.x * blockDim.x) {
        int b = pac_base(pac, (bwtint)i);
        T[i] = (uint8_t)b;
        T[2 * l_pac - 1 - i] = (uint8_t)(3 - b);     // reverse complement strand appended
    }
}
__global__ void k_count_syms(const uint8_t *T, u64 n, u64 *cnt)
{
    u64 c[4] = {0, 0, 0, 0};
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const int s = T[i];
        c[0] += s == 0; c[1] += s == 1; c[2] += s == 2; c[3] += s == 3;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u64 v = c[j];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(cnt + j, v);
    }
}
// positions of the suffixes that start with symbol c (order does not matter: they are sorted next).
// A wave counts the matches of its 4096-symbol tile first and reserves their slots with ONE atomic
// (one atomic per 64 symbols on a single counter took 1.2 s per symbol at 6.2e9 symbols).
__global__ void k_collect(const uint8_t *T, u64 n, int c, u64 *out, u64 *counter)
{
    const int lane = threadIdx.x & 63;
    const u64 lane_lt = (1ull << lane) - 1ull;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((u64)gridDim.x * blockDim.x) >> 6;
    const u64 TILE = 4096;
    for (u64 base = wave * TILE; base < n; base += n_waves * TILE) {
        unsigned int total = 0;
        for (int it = 0; it < 64; ++it) {
            const u64 i = base + (u64)it * 64 + lane;
            total += (unsigned int)__popcll(__ballot(i < n && T[i] == c));
        }
        if (total == 0) continue;
        u64 at = 0;
        if (lane == 0) at = atomicAdd(counter, (u64)total);
        at = (u64)__shfl((long long)at, 0, 64);
        for (int it = 0; it < 64; ++it) {
            const u64 i = base + (u64)it * 64 + lane;
            const bool f = i < n && T[i] == c;
            const u64 mask = __ballot(f);
            if (f) out[at + __popcll(mask & lane_lt)] = i;
            at += (u64)__popcll(mask);
        }
    }
}
// sort key of a suffix inside its chunk: the 27 symbols after the first as base-5 digits (symbol+1, 0 beyond the
// text), so a suffix that runs into '$' orders before every longer one
__global__ void k_chunk_keys(const uint8_t *T, u64 n, const u64 *pos, u64 m, u64 *key)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x) {
        const u64 p = pos[i] + 1;
        u64 k = 0;
#pragma unroll
        for (int j = 0; j < 27; ++j) k = k * 5 + (p + j < n ? (u64)T[p + j] + 1 : 0);
        key[i] = k;
    }
}
// head[i] = first element of a group of equal keys (optionally inside equal outer groups); val[i] = head ? at[i] : 0
__global__ void k_heads(const u64 *key, const u64 *outer, u64 m, u64 first_row, const u64 *rows, uint32_t *head, u64 *val)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x) {
        const bool h = i == 0 || key[i] != key[i - 1] || (outer && outer[i] != outer[i - 1]);
        head[i] = h ? 1u : 0u;
        val[i] = h ? (rows ? rows[i] : first_row + i) : 0ull;
    }
}
// after the max-scan val[i] is the rank (row of the group's first element): publish SA and inverse SA, flag ties
__global__ void k_publish(const u64 *pos, const u64 *rank, const uint32_t *head, u64 m, u64 first_row, const u64 *rows,
                          u64 *SA, u64 *ISA, uint32_t *tied)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x) {
        const u64 p = pos[i];
        SA[rows ? rows[i] : first_row + i] = p;
        ISA[p] = rank[i];
        const bool single = head[i] && (i + 1 == m || head[i + 1]);
        tied[i] = single ? 0u : 1u;
    }
}
__global__ void k_compact_tied(const uint32_t *tied, const uint32_t *off, u64 m, const u64 *pos, const u64 *rank, u64 first_row,
                               const u64 *rows, u64 *o_pos, u64 *o_grp, u64 *o_row)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x)
        if (tied[i]) { const uint32_t q = off[i]; o_pos[q] = pos[i]; o_grp[q] = rank[i]; o_row[q] = rows ? rows[i] : first_row + i; }
}
__global__ void k_next_keys(const u64 *pos, const u64 *ISA, u64 m, u64 h, u64 N, u64 *key, uint32_t *idx)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x) {
        const u64 p = pos[i] + h;
        key[i] = p < N ? ISA[p] : 0ull;
        idx[i] = (uint32_t)i;
    }
}
__global__ void k_gather64(const u64 *src, const uint32_t *idx, u64 m, u64 *dst)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
// stored BWT symbol j (the '$' row skipped) = T[SA[row]-1]
__global__ void k_bwt_syms(const u64 *sa, const uint8_t *T, u64 n, u64 primary, uint8_t *B)
{
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (u64)gridDim.x * blockDim.x) {
        u64 row = j + (j >= primary ? 1 : 0);
        B[j] = T[sa[row] - 1];
    }
}
__global__ void k_block_counts(const uint8_t *B, u64 n, uint32_t n_blocks, uint32_t *c0, uint32_t *c1, uint32_t *c2, uint32_t *c3)
{
    for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (u64)gridDim.x * blockDim.x) {
        uint32_t c[4] = {0, 0, 0, 0};
        u64 beg = b * PS_BLK_SYMS;
        for (int t = 0; t < PS_BLK_SYMS && beg + t < n; ++t) {
            int s = B[beg + t];
            c[0] += s == 0; c[1] += s == 1; c[2] += s == 2; c[3] += s == 3;
        }
        c0[b] = c[0]; c1[b] = c[1]; c2[b] = c[2]; c3[b] = c[3];
    }
}
__global__ void k_pack_blocks(const uint8_t *B, u64 n, uint32_t n_blocks, const uint32_t *c0, const uint32_t *c1,
                              const uint32_t *c2, const uint32_t *c3, OccBlock *blocks)
{
    for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (u64)gridDim.x * blockDim.x) {
        u64 beg = b * PS_BLK_SYMS;
        int m = beg < n ? (int)(n - beg < (u64)PS_BLK_SYMS ? n - beg : (u64)PS_BLK_SYMS) : 0;
        uint32_t cnt[4] = {c0[b], c1[b], c2[b], c3[b]};
        OccBlock blk;
        blk_pack(blk, B + (beg < n ? beg : 0), m, cnt);
        blocks[b] = blk;
    }
}
// sampled SA: low words, then one word of bit-32 flags per 32 samples
__global__ void k_sample_sa(const u64 *sa, uint32_t n_sa, int intv, uint32_t *lo, uint32_t *hi)
{
    const uint32_t n_words = (n_sa + 31) / 32;
    for (u64 wd = (u64)blockIdx.x * blockDim.x + threadIdx.x; wd < n_words; wd += (u64)gridDim.x * blockDim.x) {
        uint32_t bits = 0;
        for (int j = 0; j < 32; ++j) {
            const u64 t = wd * 32 + j;
            if (t >= n_sa) break;
            const u64 v = t == 0 ? ~0ull : sa[t * (u64)intv];
            lo[t] = (uint32_t)v;
            bits |= (uint32_t)((v >> 32) & 1ull) << j;
        }
        hi[wd] = bits;
    }
}

void Index::refresh_view()
{
    view.blocks = blocks.p; view.sa = sa.p; view.sa_hi = sa.p + view.n_sa; view.pac = pac.p;
    view.l_pac = (bwtint)ref.l_pac; view.seq_len = (bwtint)(2 * ref.l_pac);
    view.n_blocks = (uint32_t)blocks.n; view.sa_intv = 32;
    view.jump = jump_levels > 0 ? jump.p : nullptr; view.jump_levels = jump_levels;
}

void index_build_jump(Index &ix, hipStream_t s)
{
    int levels = jump_levels_for(ix.view.seq_len);
    if (const char *e = std::getenv("PS_JUMP_LEVELS")) levels = std::max(0, std::min(PS_JUMP_MAX_LEVELS, std::atoi(e)));
    ix.jump_levels = 0; ix.view.jump = nullptr; ix.view.jump_levels = 0;
    if (levels == 0) return;
    ix.jump.alloc(jump_words(levels));
    launch_jump_build(ix.view, ix.jump.p, levels, s);
    PS_HIP(hipStreamSynchronize(s));
    ix.jump_levels = levels; ix.view.jump = ix.jump.p; ix.view.jump_levels = levels;
}

namespace {
struct Sorter {          // hipcub calls with one grow-only temporary buffer
    DevBuf<uint8_t> tmp; hipStream_t s;
    void need(size_t b) { if (b > tmp.n) tmp.alloc(b + b / 8 + 256); }
    template <class V> void pairs(const u64 *kin, u64 *kout, const V *vin, V *vout, size_t m, int end_bit)
    {
        size_t b = 0;
        PS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, b, kin, kout, vin, vout, m, 0, end_bit, s));
        need(b); b = tmp.n;
        PS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, b, kin, kout, vin, vout, m, 0, end_bit, s));
    }
    void max_scan(u64 *v, size_t m)
    {
        size_t b = 0;
        PS_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, b, v, v, hipcub::Max(), m, s));
        need(b); b = tmp.n;
        PS_HIP(hipcub::DeviceScan::InclusiveScan(tmp.p, b, v, v, hipcub::Max(), m, s));
    }
    void ex_sum(const uint32_t *in, uint32_t *out, size_t m)
    {
        size_t b = 0;
        PS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, b, in, out, m, s));
        need(b); b = tmp.n;
        PS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, b, in, out, m, s));
    }
};
struct Tied { DevBuf<u64> pos, grp, row; size_t m = 0; };   // suffixes still in groups of two or more, in row order

// ranks are in `rank` (after the max-scan), heads in `head`: publish, then append the tied ones to `out`
void publish_and_compact(Sorter &so, const u64 *pos, u64 *rank, const uint32_t *head, size_t m, u64 first_row, const u64 *rows,
                         u64 *SA, u64 *ISA, Tied &out)
{
    hipStream_t s = so.s;
    DevBuf<uint32_t> tied, off; tied.alloc(m); off.alloc(m);
    hipLaunchKernelGGL(k_publish, dim3(GRID), dim3(BLK), 0, s, pos, rank, head, (u64)m, first_row, rows, SA, ISA, tied.p);
    so.ex_sum(tied.p, off.p, m);
    uint32_t last_off = 0, last_flag = 0;
    PS_HIP(hipMemcpyAsync(&last_off, off.p + (m - 1), 4, hipMemcpyDeviceToHost, s));
    PS_HIP(hipMemcpyAsync(&last_flag, tied.p + (m - 1), 4, hipMemcpyDeviceToHost, s));
    PS_HIP(hipStreamSynchronize(s));
    out.m = (size_t)last_off + last_flag;
    if (out.m) {
        out.pos.alloc(out.m); out.grp.alloc(out.m); out.row.alloc(out.m);
        hipLaunchKernelGGL(k_compact_tied, dim3(GRID), dim3(BLK), 0, s, tied.p, off.p, (u64)m, pos, rank, first_row, rows, out.pos.p, out.grp.p, out.row.p);
    }
    PS_HIP(hipStreamSynchronize(s));
}
}  // namespace

void index_build(const char *fa, Index &ix, hipStream_t s)
{
    auto t0 = std::chrono::steady_clock::now();
    load_fasta(fa, ix.ref);
    const u64 l_pac = (u64)ix.ref.l_pac, n = 2 * l_pac, N = n + 1;
    if (N >= PS_MAX_ROWS) throw Error("reference too large: forward + reverse strand must stay below 2^33 rows (4.29 Gbp)");
    ix.pac.alloc(ix.ref.pac.size());
    ix.pac.upload(ix.ref.pac.data(), ix.ref.pac.size(), s);
    DevBuf<uint8_t> T; T.alloc(n + 64);
    hipLaunchKernelGGL(k_expand_text, dim3(GRID), dim3(BLK), 0, s, ix.pac.p, T.p, l_pac);
    DevBuf<u64> dcnt; dcnt.alloc(5); dcnt.zero(s);
    hipLaunchKernelGGL(k_count_syms, dim3(GRID), dim3(BLK), 0, s, T.p, n, dcnt.p);
    u64 cnt[5] = {0, 0, 0, 0, 0};
    dcnt.download(cnt, 5, s);
    PS_HIP(hipStreamSynchronize(s));
    for (int c = 0; c < 4; ++c)
        if (cnt[c] >= 0xFFFFFF00ull) throw Error("reference too skewed: one base occurs 2^32 times or more");
    DevBuf<u64> SA, ISA; SA.alloc(N); ISA.alloc(N);
    Sorter so; so.s = s;
    // ---- round 0: chunk by first symbol, order by the next 27 ----
    {
        const u64 nn = n;                                  // row 0: the empty suffix
        PS_HIP(hipMemcpyAsync(SA.p, &nn, 8, hipMemcpyHostToDevice, s));
        const u64 zero = 0;
        PS_HIP(hipMemcpyAsync(ISA.p + n, &zero, 8, hipMemcpyHostToDevice, s));
        PS_HIP(hipStreamSynchronize(s));
    }
    std::vector<Tied> parts(4);
    u64 first_row = 1;
    for (int c = 0; c < 4; ++c) {
        const size_t m = (size_t)cnt[c];
        if (m == 0) continue;
        DevBuf<u64> posA, posB, keyA, keyB; posA.alloc(m); posB.alloc(m); keyA.alloc(m); keyB.alloc(m);
        dcnt.zero(s);
        hipLaunchKernelGGL(k_collect, dim3(GRID), dim3(BLK), 0, s, T.p, n, c, posA.p, dcnt.p + 4);
        hipLaunchKernelGGL(k_chunk_keys, dim3(GRID), dim3(BLK), 0, s, T.p, n, posA.p, (u64)m, keyA.p);
        so.pairs<u64>(keyA.p, keyB.p, posA.p, posB.p, m, 63);
        posA.release();
        DevBuf<uint32_t> head; head.alloc(m);
        u64 *rank = keyA.p;                                // reuse: the unsorted keys are no longer needed
        hipLaunchKernelGGL(k_heads, dim3(GRID), dim3(BLK), 0, s, keyB.p, (const u64 *)nullptr, (u64)m, first_row, (const u64 *)nullptr, head.p, rank);
        so.max_scan(rank, m);
        publish_and_compact(so, posB.p, rank, head.p, m, first_row, nullptr, SA.p, ISA.p, parts[c]);
        first_row += m;
    }
    Tied cur;
    for (int c = 0; c < 4; ++c) cur.m += parts[c].m;
    if (cur.m) {
        cur.pos.alloc(cur.m); cur.grp.alloc(cur.m); cur.row.alloc(cur.m);
        size_t at = 0;
        for (int c = 0; c < 4; ++c) {
            if (!parts[c].m) continue;
            PS_HIP(hipMemcpyAsync(cur.pos.p + at, parts[c].pos.p, parts[c].m * 8, hipMemcpyDeviceToDevice, s));
            PS_HIP(hipMemcpyAsync(cur.grp.p + at, parts[c].grp.p, parts[c].m * 8, hipMemcpyDeviceToDevice, s));
            PS_HIP(hipMemcpyAsync(cur.row.p + at, parts[c].row.p, parts[c].m * 8, hipMemcpyDeviceToDevice, s));
            at += parts[c].m;
        }
        PS_HIP(hipStreamSynchronize(s));
    }
    parts.clear();
    ix.sa_rounds = 1;
    // ---- doubling rounds over what is still tied ----
    for (u64 h = 28; cur.m; h *= 2) {
        if (h > 2 * N) throw Error("suffix sorting did not converge");
        if (cur.m >= 0xFFFFFFFFull) throw Error("reference too repetitive for the 32-bit tie lists of the index builder");
        const size_t m = cur.m;
        DevBuf<u64> key, keyS, g1, g1S, pos2; DevBuf<uint32_t> idx0, idx1, idx2, head;
        key.alloc(m); keyS.alloc(m); idx0.alloc(m); idx1.alloc(m);
        hipLaunchKernelGGL(k_next_keys, dim3(GRID), dim3(BLK), 0, s, cur.pos.p, ISA.p, (u64)m, h, N, key.p, idx0.p);
        so.pairs<uint32_t>(key.p, keyS.p, idx0.p, idx1.p, m, 34);
        g1.alloc(m); g1S.alloc(m); idx2.alloc(m);
        hipLaunchKernelGGL(k_gather64, dim3(GRID), dim3(BLK), 0, s, cur.grp.p, idx1.p, (u64)m, g1.p);
        so.pairs<uint32_t>(g1.p, g1S.p, idx1.p, idx2.p, m, 34);     // stable: ties keep the order of the first sort
        // g1S equals cur.grp (it was already in row order); element i now belongs in row cur.row[i]
        pos2.alloc(m);
        hipLaunchKernelGGL(k_gather64, dim3(GRID), dim3(BLK), 0, s, cur.pos.p, idx2.p, (u64)m, pos2.p);
        hipLaunchKernelGGL(k_gather64, dim3(GRID), dim3(BLK), 0, s, key.p, idx2.p, (u64)m, keyS.p);
        head.alloc(m);
        u64 *rank = g1.p;
        hipLaunchKernelGGL(k_heads, dim3(GRID), dim3(BLK), 0, s, keyS.p, g1S.p, (u64)m, 0ull, cur.row.p, head.p, rank);
        so.max_scan(rank, m);
        Tied next;
        publish_and_compact(so, pos2.p, rank, head.p, m, 0, cur.row.p, SA.p, ISA.p, next);
        cur = std::move(next);
        ++ix.sa_rounds;
    }
    u64 primary = 0;
    PS_HIP(hipMemcpyAsync(&primary, ISA.p, 8, hipMemcpyDeviceToHost, s));   // row of the whole text = where '$' sits in the last column
    PS_HIP(hipStreamSynchronize(s));
    ISA.release(); so.tmp.release();
    // BWT, Occ blocks, sampled SA
    DevBuf<uint8_t> B; B.alloc(n + 1);
    hipLaunchKernelGGL(k_bwt_syms, dim3(GRID), dim3(BLK), 0, s, SA.p, T.p, n, primary, B.p);
    const uint32_t n_blocks = (uint32_t)(n / PS_BLK_SYMS + 1);
    DevBuf<uint32_t> c[4], cs[4];
    for (int j = 0; j < 4; ++j) { c[j].alloc(n_blocks); cs[j].alloc(n_blocks); }
    hipLaunchKernelGGL(k_block_counts, dim3(GRID), dim3(BLK), 0, s, B.p, n, n_blocks, c[0].p, c[1].p, c[2].p, c[3].p);
    uint32_t last_c[4], last_s[4];
    for (int j = 0; j < 4; ++j) {
        so.ex_sum(c[j].p, cs[j].p, n_blocks);
        PS_HIP(hipMemcpyAsync(&last_c[j], c[j].p + (n_blocks - 1), 4, hipMemcpyDeviceToHost, s));
        PS_HIP(hipMemcpyAsync(&last_s[j], cs[j].p + (n_blocks - 1), 4, hipMemcpyDeviceToHost, s));
    }
    ix.blocks.alloc(n_blocks);
    hipLaunchKernelGGL(k_pack_blocks, dim3(GRID), dim3(BLK), 0, s, B.p, n, n_blocks, cs[0].p, cs[1].p, cs[2].p, cs[3].p, ix.blocks.p);
    const uint32_t n_sa = (uint32_t)((n + 32) / 32);
    ix.sa.alloc(Index::sa_words(n_sa));
    hipLaunchKernelGGL(k_sample_sa, dim3(GRID), dim3(BLK), 0, s, SA.p, n_sa, 32, ix.sa.p, ix.sa.p + n_sa);
    PS_HIP(hipStreamSynchronize(s));
    ix.view.primary = primary;
    ix.view.L2[0] = 0;
    for (int j = 0; j < 4; ++j) ix.view.L2[j + 1] = ix.view.L2[j] + last_c[j] + last_s[j];
    ix.view.n_sa = n_sa;
    ix.refresh_view();
    if (ix.view.L2[4] != (bwtint)n) throw Error("index build: symbol counts do not add up");
    for (int j = 0; j < 4; ++j)
        if (ix.view.L2[j + 1] - ix.view.L2[j] != cnt[j]) throw Error("index build: BWT symbol counts differ from the text's");
    index_build_jump(ix, s);
    ix.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// ------------------------------------------------------------- files ---------
// <ref>.bwt (header + Occ blocks), <ref>.sa, <ref>.pac, <ref>.ann (contigs + holes), this project's formats.
static const char MAGIC_BWT[8] = {'P', 'S', 'B', 'W', 'T', '0', '2', 0};

std::string index_meta_serialize(const Index &ix)
{
    std::ostringstream o;
    o << "PSANN02\n" << ix.ref.l_pac << ' ' << ix.ref.contigs.size() << ' ' << ix.ref.holes.size() << '\n';
    o << ix.view.seq_len << ' ' << ix.view.primary << ' ' << ix.view.L2[0] << ' ' << ix.view.L2[1] << ' ' << ix.view.L2[2] << ' '
      << ix.view.L2[3] << ' ' << ix.view.L2[4] << ' ' << ix.blocks.n << ' ' << ix.view.n_sa << ' ' << ix.pac.n << '\n';
    for (const Contig &c : ix.ref.contigs) o << c.name << '\t' << c.offset << '\t' << c.len << '\t' << c.n_ambs << '\t' << c.anno << '\n';
    for (const Hole &h : ix.ref.holes) o << h.offset << ' ' << h.len << ' ' << (int)(unsigned char)h.amb << '\n';
    return o.str();
}
void index_meta_deserialize(const std::string &blob, Index &ix)
{
    std::istringstream in(blob);
    std::string magic; std::getline(in, magic);
    if (magic != "PSANN02") throw Error("bad index metadata");
    size_t nc, nh, nb, ns, np; uint64_t v[7];
    in >> ix.ref.l_pac >> nc >> nh;
    for (int j = 0; j < 7; ++j) in >> v[j];
    in >> nb >> ns >> np;
    ix.view.seq_len = (bwtint)v[0]; ix.view.primary = (bwtint)v[1];
    for (int j = 0; j < 5; ++j) ix.view.L2[j] = (bwtint)v[2 + j];
    std::string line; std::getline(in, line);
    ix.ref.contigs.clear(); ix.ref.holes.clear();
    for (size_t i = 0; i < nc; ++i) {
        std::getline(in, line);
        Contig c; size_t a = line.find('\t'), b = line.find('\t', a + 1), d = line.find('\t', b + 1), e = line.find('\t', d + 1);
        c.name = line.substr(0, a); c.offset = std::stoll(line.substr(a + 1, b - a - 1));
        c.len = std::stoi(line.substr(b + 1, d - b - 1)); c.n_ambs = std::stoi(line.substr(d + 1, e - d - 1)); c.anno = line.substr(e + 1);
        ix.ref.contigs.push_back(c);
    }
    for (size_t i = 0; i < nh; ++i) { Hole h; int amb; in >> h.offset >> h.len >> amb; h.amb = (char)amb; ix.ref.holes.push_back(h); }
    ix.view.n_blocks = (uint32_t)nb; ix.view.n_sa = (uint32_t)ns; ix.view.sa_intv = 32; ix.view.l_pac = (bwtint)ix.ref.l_pac;
}

bool index_files_exist(const std::string &prefix)
{
    for (const char *ext : {".bwt", ".sa", ".pac", ".ann"}) { std::ifstream f(prefix + ext); if (!f.good()) return false; }
    return true;
}

template <class T> static void write_dev(const std::string &path, const char *magic8, const DevBuf<T> &d)
{
    std::vector<T> h(d.n);
    PS_HIP(hipMemcpy(h.data(), d.p, d.n * sizeof(T), hipMemcpyDeviceToHost));
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error("cannot write " + path);
    uint64_t cnt = d.n;
    bool ok = std::fwrite(magic8, 1, 8, f) == 8 && std::fwrite(&cnt, 8, 1, f) == 1 && (d.n == 0 || std::fwrite(h.data(), sizeof(T), d.n, f) == d.n);
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) throw Error("short write on " + path);
}
// file -> device: a few reader threads, each with two pinned staging buffers, take 32-MB pieces of the file in turn (pread) and
// send every piece up as soon as it is in -- the next one is read while the last is on its way.  One thread read ~12 GB/s out of the
// page cache (0.37 s for the 4.7 GB of an hg19-size index, all of it in front of ps_map's first search launch); the bus takes four
// times that.  (A pageable 3.6 GB std::vector first cost 1.4 s.)  The staging buffers come from the pinned-buffer cache of
// ps_pipeline.hip and are shared by the three files of a load.
void *pin_cache_take(size_t need, size_t &got);
void pin_cache_give(void *p, size_t bytes);
namespace {
struct LoadStage {
    static const size_t PIECE = (size_t)32 << 20;
    int n_thr = 4, device = 0;
    std::vector<void *> buf; std::vector<size_t> got; std::vector<hipEvent_t> done;
    LoadStage()
    {
        if (const char *e = std::getenv("PS_LOAD_THREADS")) n_thr = std::max(1, std::min(16, std::atoi(e)));
        PS_HIP(hipGetDevice(&device));
        buf.assign((size_t)2 * n_thr, nullptr); got.assign((size_t)2 * n_thr, 0); done.assign((size_t)2 * n_thr, nullptr);
    }
    void ready(int slot)                              // buffers are taken by the thread that first needs them (a small file uses one or two)
    {
        if (!buf[slot]) { buf[slot] = pin_cache_take(PIECE, got[slot]); PS_HIP(hipEventCreateWithFlags(&done[slot], hipEventDisableTiming)); }
    }
    ~LoadStage() { for (size_t k = 0; k < buf.size(); ++k) { if (done[k]) (void)hipEventDestroy(done[k]); if (buf[k]) pin_cache_give(buf[k], got[k]); } }
};
}
template <class T> static void read_dev(LoadStage &st, const std::string &path, const char *magic8, DevBuf<T> &d, hipStream_t s, std::vector<uint8_t> *host_copy = nullptr)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error("cannot open " + path);
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    char head[16]; uint64_t cnt = 0;
    if (::pread(fd, head, 16, 0) != 16 || std::memcmp(head, magic8, 8) != 0) throw Error("bad header in " + path);
    std::memcpy(&cnt, head + 8, 8);
    d.alloc(cnt);
    const size_t total = (size_t)cnt * sizeof(T), PIECE = LoadStage::PIECE, n_pieces = (total + PIECE - 1) / PIECE;
    if (host_copy) host_copy->resize(total);
    std::atomic<size_t> next{0}; std::atomic<bool> ok{true};
    std::mutex err_mu; std::string err;
    auto reader = [&](int t) {
        try {
            PS_HIP(hipSetDevice(st.device));
            int k = 0; bool used[2] = {false, false};
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n_pieces || !ok) break;
                const int slot = 2 * t + k;
                st.ready(slot);
                if (used[k]) PS_HIP(hipEventSynchronize(st.done[slot]));      // the copy that last used this buffer has finished
                const size_t at = i * PIECE, m = std::min(PIECE, total - at);
                size_t have = 0;
                while (have < m) { const ssize_t r = ::pread(fd, (char *)st.buf[slot] + have, m - have, (off_t)(16 + at + have)); if (r <= 0) break; have += (size_t)r; }
                if (have != m) { ok = false; break; }
                if (host_copy) std::memcpy(host_copy->data() + at, st.buf[slot], m);
                PS_HIP(hipMemcpyAsync(reinterpret_cast<uint8_t *>(d.p) + at, st.buf[slot], m, hipMemcpyHostToDevice, s));
                PS_HIP(hipEventRecord(st.done[slot], s)); used[k] = true;
                k ^= 1;
            }
            for (int q = 0; q < 2; ++q) if (used[q]) PS_HIP(hipEventSynchronize(st.done[2 * t + q]));
        } catch (const std::exception &e) { std::lock_guard<std::mutex> l(err_mu); if (err.empty()) err = e.what(); ok = false; }
    };
    {
        const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)st.n_thr, n_pieces));
        std::vector<std::thread> th; for (int t = 1; t < nt; ++t) th.emplace_back(reader, t);
        reader(0);
        for (auto &x : th) x.join();
    }
    PS_HIP(hipStreamSynchronize(s));
    if (!err.empty()) throw Error(err);
    if (!ok) throw Error("truncated " + path);
}

void index_save(const Index &ix, const std::string &prefix)
{
    write_dev(prefix + ".bwt", MAGIC_BWT, ix.blocks);
    write_dev(prefix + ".sa", "PSSA0002", ix.sa);
    write_dev(prefix + ".pac", "PSPAC001", ix.pac);
    std::ofstream a(prefix + ".ann", std::ios::binary);
    a << index_meta_serialize(ix);
    if (!a.good()) throw Error("cannot write " + prefix + ".ann");
}

// the .pac file mapped for the host side; if it cannot be mapped, read into ref.pac
static void map_pac(const std::string &path, size_t n_bytes, RefSeq &ref)
{
    ref.pac_map.reset(); ref.pac_view = nullptr;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error("cannot open " + path);
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    const size_t total = 16 + n_bytes;
    void *m = ::mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
    if (m != MAP_FAILED) {
        ref.pac_map = std::shared_ptr<const void>(m, [total](const void *q) { ::munmap(const_cast<void *>(q), total); });
        ref.pac_view = static_cast<const uint8_t *>(m) + 16;
        ref.pac.clear(); ref.pac.shrink_to_fit();
        return;
    }
    ref.pac.resize(n_bytes);
    size_t have = 0;
    while (have < n_bytes) { const ssize_t r = ::pread(fd, ref.pac.data() + have, n_bytes - have, (off_t)(16 + have)); if (r <= 0) throw Error("truncated " + path); have += (size_t)r; }
}

void index_load(const std::string &prefix, Index &ix, hipStream_t s)
{
    std::ifstream a(prefix + ".ann", std::ios::binary);
    if (!a.good()) throw Error("cannot open " + prefix + ".ann (run ps_index first)");
    std::stringstream ss; ss << a.rdbuf();
    index_meta_deserialize(ss.str(), ix);
    LoadStage st;
    read_dev(st, prefix + ".bwt", MAGIC_BWT, ix.blocks, s);
    read_dev(st, prefix + ".sa", "PSSA0002", ix.sa, s);
    read_dev(st, prefix + ".pac", "PSPAC001", ix.pac, s, nullptr);
    map_pac(prefix + ".pac", ix.pac.n, ix.ref);                             // the host reads the pac too (MD tags)
    bwtint primary = ix.view.primary; bwtint L2[5]; std::memcpy(L2, ix.view.L2, sizeof L2);
    ix.refresh_view();
    ix.view.primary = primary; std::memcpy(ix.view.L2, L2, sizeof L2);
    index_build_jump(ix, s);
}

// only what the error-profile stage needs of an index: contigs, holes and the packed forward strand on the device
void index_load_pac(const std::string &prefix, Index &ix, hipStream_t s)
{
    std::ifstream a(prefix + ".ann", std::ios::binary);
    if (!a.good()) throw Error("cannot open " + prefix + ".ann (run ps_index first)");
    std::stringstream ss; ss << a.rdbuf();
    index_meta_deserialize(ss.str(), ix);
    LoadStage st;
    read_dev(st, prefix + ".pac", "PSPAC001", ix.pac, s, nullptr);
    ix.view.pac = ix.pac.p; ix.view.l_pac = (bwtint)ix.ref.l_pac;
}

// The index of one device copied to another: its blobs (and the jump table) over xGMI, no file read (ps_map with several devices loads the
// files once).  Falls back to a copy staged by the runtime when the devices cannot reach each other directly.
void index_clone(const Index &src, int src_device, Index &dst, int dst_device, hipStream_t s)
{
    dst.ref = src.ref; dst.build_ms = src.build_ms; dst.sa_rounds = src.sa_rounds;
    int can = 0;
    if (src_device != dst_device && hipDeviceCanAccessPeer(&can, dst_device, src_device) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(src_device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
    }
    dst.blocks.alloc(src.blocks.n); dst.sa.alloc(src.sa.n); dst.pac.alloc(src.pac.n);
    dst.jump_levels = src.jump_levels;
    if (src.jump_levels) { dst.jump.alloc(src.jump.n); PS_HIP(hipMemcpyPeerAsync(dst.jump.p, dst_device, src.jump.p, src_device, src.jump.n * sizeof(uint32_t), s)); }
    PS_HIP(hipMemcpyPeerAsync(dst.blocks.p, dst_device, src.blocks.p, src_device, src.blocks.n * sizeof(OccBlock), s));
    PS_HIP(hipMemcpyPeerAsync(dst.sa.p, dst_device, src.sa.p, src_device, src.sa.n * sizeof(uint32_t), s));
    PS_HIP(hipMemcpyPeerAsync(dst.pac.p, dst_device, src.pac.p, src_device, src.pac.n, s));
    PS_HIP(hipStreamSynchronize(s));
    const bwtint primary = src.view.primary; bwtint L2[5]; std::memcpy(L2, src.view.L2, sizeof L2);
    dst.view = src.view;
    dst.refresh_view();
    dst.view.primary = primary; std::memcpy(dst.view.L2, L2, sizeof L2);
}

}  // namespace ps
