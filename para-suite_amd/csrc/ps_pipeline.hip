// ps_pipeline.hip -- host orchestration of one mapping job.
//
// Replaces what the reference runs as two child processes,
//   bwa parasuite|aln ... -f P.sai     PARAsuiteMapping.java:63-77 / BWAMapping.java:51-61
//   bwa samse ref P.sai fq -f P.sam    PARAsuiteMapping.java:85-92 / BWAMapping.java:68-75
// by one pass: reads binned by length and 2-bit packed in HBM -> width kernel ->
// backtracking kernel -> tie-break selection (one drand48 stream in input order)
// -> SA-walk kernel -> banded-DP kernel for gapped hits -> SAM text.
// No stage has a CPU implementation of the kernels' work: without a HIP device
// every entry point fails.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <chrono>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include "ps_pipeline.h"
#include "ps_core.h"

namespace ps {

typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

void require_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) throw Error("no HIP device available: parasuite-hip has no CPU path");
    if (device < 0 || device >= n) throw Error("HIP device index out of range");
    PS_HIP(hipSetDevice(device));
}

Ctx::~Ctx() { if (stream) (void)hipStreamDestroy(stream); }

// ------------------------------------------------------------- read input ----
static inline uint8_t code_of(int ch)
{
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                  case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

// FASTQ / FASTA; read name = header up to the first white space, a trailing /1 or /2 removed
void load_reads(const char *path, ReadSet &rs)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) throw Error(std::string("cannot open reads ") + path);
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)sz + 1);
    if (sz && std::fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); throw Error(std::string("short read on ") + path); }
    std::fclose(f);
    const size_t n = (size_t)sz;
    rs = ReadSet();
    rs.off.push_back(0); rs.name_off.push_back(0);
    size_t i = 0;
    bool any_qual = false;
    while (i < n) {
        while (i < n && buf[i] != '@' && buf[i] != '>') ++i;
        if (i >= n) break;
        const bool fq = buf[i] == '@';
        size_t s = ++i;
        while (i < n && !std::isspace((unsigned char)buf[i])) ++i;
        size_t nl = i - s;
        if (nl > 2 && buf[s + nl - 2] == '/' && (buf[s + nl - 1] == '1' || buf[s + nl - 1] == '2')) nl -= 2;
        rs.names.insert(rs.names.end(), buf.data() + s, buf.data() + s + nl);
        rs.name_off.push_back((int64_t)rs.names.size());
        while (i < n && buf[i] != '\n') ++i;
        ++i;
        size_t before = rs.seq.size();
        while (i < n && buf[i] != (fq ? '+' : '>')) { if (std::isgraph((unsigned char)buf[i])) rs.seq.push_back(code_of(buf[i])); ++i; }
        int32_t len = (int32_t)(rs.seq.size() - before);
        rs.len.push_back(len);
        rs.off.push_back((int64_t)rs.seq.size());
        size_t qbefore = rs.qual.size();
        if (fq && i < n) {
            while (i < n && buf[i] != '\n') ++i;
            ++i;
            while (i < n && (int32_t)(rs.qual.size() - qbefore) < len) { if (std::isgraph((unsigned char)buf[i])) rs.qual.push_back(buf[i]); ++i; }
            any_qual = true;
        }
        rs.qual.resize(qbefore + (size_t)len, '!');
        ++rs.n;
    }
    rs.has_qual = any_qual;
}

void reads_from_codes(int64_t n, int len, const uint8_t *codes, ReadSet &rs)
{
    rs = ReadSet();
    rs.n = n; rs.len.assign((size_t)n, len); rs.off.resize((size_t)n + 1); rs.name_off.resize((size_t)n + 1);
    rs.seq.assign(codes, codes + (size_t)n * len);
    for (auto &c : rs.seq) if (c > 4) c = 4;
    char nm[32];
    rs.name_off[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        rs.off[i] = i * len;
        int l = std::snprintf(nm, sizeof nm, "r%lld", (long long)i);
        rs.names.insert(rs.names.end(), nm, nm + l);
        rs.name_off[i + 1] = (int64_t)rs.names.size();
    }
    rs.off[n] = n * (int64_t)len;
    rs.has_qual = false;
}

// ------------------------------------------------------------ batch set-up ---
std::unique_ptr<Batch> batch_create(Ctx *ctx, ReadSet &&rs_in)
{
    std::unique_ptr<Batch> b(new Batch());
    b->ctx = ctx; b->rs = std::move(rs_in);
    const ReadSet &rs = b->rs;
    std::map<int, int> bin_of_len;
    b->read_bin.resize((size_t)rs.n); b->read_local.resize((size_t)rs.n);
    for (int64_t g = 0; g < rs.n; ++g) {
        int len = rs.len[g];
        if (len < 1) throw Error("empty read in input");
        auto it = bin_of_len.find(len);
        if (it == bin_of_len.end()) { it = bin_of_len.emplace(len, (int)b->bins.size()).first; b->bins.emplace_back(); b->bins.back().len = len; }
        Bin &bin = b->bins[it->second];
        b->read_bin[g] = it->second; b->read_local[g] = (int32_t)bin.ids.size();
        bin.ids.push_back((int32_t)g);
    }
    for (Bin &bin : b->bins) {
        std::string err;
        if (!make_model(ctx->opt, bin.len, bin.md, err)) throw Error(err);
        const size_t n = bin.ids.size();
        bin.n_bw = (bin.len + 15) / 16; bin.n_mw = (bin.len + 31) / 32;
        bin.h_bases.assign((size_t)bin.n_bw * n, 0); bin.h_nmask.assign((size_t)bin.n_mw * n, 0);
        for (size_t r = 0; r < n; ++r) {
            const uint8_t *s = rs.seq.data() + rs.off[bin.ids[r]];
            for (int j = 0; j < bin.len; ++j) {
                if (s[j] > 3) bin.h_nmask[(size_t)(j >> 5) * n + r] |= 1u << (j & 31);
                else bin.h_bases[(size_t)(j >> 4) * n + r] |= (uint32_t)s[j] << (2 * (j & 15));
            }
        }
        bin.bases.alloc(bin.h_bases.size()); bin.nmask.alloc(bin.h_nmask.size());
        bin.bases.upload(bin.h_bases.data(), bin.h_bases.size(), ctx->stream);
        bin.nmask.upload(bin.h_nmask.data(), bin.h_nmask.size(), ctx->stream);
    }
    b->d_stats.alloc(3);
    PS_HIP(hipStreamSynchronize(ctx->stream));
    return b;
}

// --------------------------------------------------------------- search -------
__global__ void k_gather_alns(const AlnRec *alns, int aln_cap, const int32_t *n_aln, const uint32_t *off, int n, AlnRec *out)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        int m = n_aln[r]; if (m > aln_cap) m = aln_cap;
        for (int j = 0; j < m; ++j) out[off[r] + j] = alns[(size_t)r * aln_cap + j];
    }
}
__global__ void k_clip_counts(const int32_t *n_aln, int aln_cap, int n, uint32_t *out)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) { int m = n_aln[r]; out[r] = (uint32_t)(m > aln_cap ? aln_cap : (m < 0 ? 0 : m)); }
}

struct EvTimer {
    hipEvent_t a, b; hipStream_t s;
    explicit EvTimer(hipStream_t st) : s(st) { PS_HIP(hipEventCreate(&a)); PS_HIP(hipEventCreate(&b)); PS_HIP(hipEventRecord(a, s)); }
    double stop() { PS_HIP(hipEventRecord(b, s)); PS_HIP(hipEventSynchronize(b)); float ms = 0; PS_HIP(hipEventElapsedTime(&ms, a, b)); return ms; }
    ~EvTimer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
};

// results of one search launch; the buffers are page-locked and owned by the caller
struct SearchOut { PinBuf *pn, *ps, *po, *pa; int32_t *n_aln; uint8_t *status; uint32_t *off; AlnRec *alns; };

// width + backtracking kernels over n reads of one length that are already packed on the device
static void run_search(Batch &b, const Model &md, int n, const uint32_t *d_bases, const uint32_t *d_nmask,
                       uint32_t pool_cap, int aln_cap, SearchOut &out)
{
    Ctx *ctx = b.ctx; hipStream_t s = ctx->stream;
    const int len = md.len, seed_len = md.seed_len;
    uint32_t *w = ctx->ws_get<uint32_t>("w", (size_t)(len + 1) * n);
    uint32_t *cwb = ctx->ws_get<uint32_t>("cwb", (size_t)lm_ncw(len) * n);
    uint32_t *cswb = ctx->ws_get<uint32_t>("cswb", (size_t)(lm_ncsw(seed_len) + 1) * n);
    uint8_t *status = ctx->ws_get<uint8_t>("status", n);
    AlnRec *alns = ctx->ws_get<AlnRec>("alns", (size_t)n * aln_cap);
    int32_t *n_aln = ctx->ws_get<int32_t>("n_aln", n);
    WidthArgs wa;
    wa.ix = ctx->ix.view; wa.n_reads = n; wa.len = len; wa.seed_len = seed_len; wa.use_seed = md.use_seed;
    wa.bases = d_bases; wa.nmask = d_nmask; wa.w = w; wa.cwb = cwb; wa.cswb = cswb; wa.stats = b.d_stats.p + 0;
    { EvTimer t(s); launch_width(wa, s); PS_HIP(hipGetLastError()); b.tm.ms_width += t.stop(); ++b.tm.n_width_launches; }
    int dev_cus = 256;
    { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, ctx->device) == hipSuccess && p.multiProcessorCount > 0) dev_cus = p.multiProcessorCount; }
    const bool wide = pool_cap > 65535;                       // narrow entries link with 16-bit indices
    const int lm = lm_bytes(len, seed_len, md.n_buckets, wide);
    int per_cu = (int)((size_t)(160 * 1024) / ((size_t)256 * lm));
    if (per_cu < 1) throw Error("read length / score range too large for the per-lane LDS state");
    if (per_cu > 4) per_cu = 4;
    int blocks = ctx->bt_blocks > 0 ? ctx->bt_blocks : dev_cus * per_cu;
    int need = (n + 255) / 256;
    if (blocks > need) blocks = need;
    // bound the lanes by stack memory (the widest tier keeps 64 MB per lane)
    const size_t per_lane = (size_t)pool_cap * (wide ? sizeof(Entry) : 16) + (wide ? PS_MAX_BUCKETS * 4 : 0);
    const size_t max_lanes = ((size_t)(wide ? 32 : 64) << 30) / per_lane;
    if ((size_t)blocks * 256 > max_lanes) blocks = (int)std::max<size_t>(1, max_lanes / 256);
    const int n_lanes = blocks * 256;
    uint8_t *pool = ctx->ws_get<uint8_t>("pool", (size_t)n_lanes * pool_cap * (wide ? sizeof(Entry) : 16));
    uint32_t *heads = wide ? ctx->ws_get<uint32_t>("heads", (size_t)n_lanes * PS_MAX_BUCKETS) : nullptr;
    uint32_t *queue = ctx->ws_get<uint32_t>("queue", 16);
    PS_HIP(hipMemsetAsync(queue, 0, 64, s));
    BtArgs a; std::memset(&a, 0, sizeof a);
    a.ix = ctx->ix.view; a.md = md; a.n_reads = n; a.len = len; a.n_lanes = n_lanes;
    a.bases = d_bases; a.nmask = d_nmask; a.n_bw = (len + 15) / 16; a.n_mw = (len + 31) / 32;
    a.w = w; a.cwb = cwb; a.cswb = cswb;
    a.alns = alns; a.aln_cap = aln_cap; a.n_aln = n_aln; a.status = status;
    a.pool = pool; a.pool_cap = pool_cap; a.heads = heads; a.wide = wide ? 1 : 0; a.stats = b.d_stats.p + 1;
    a.queue = queue; a.fetch_min = ctx->fetch_min; a.hit_min = ctx->hit_min;
    uint32_t *riters = nullptr;
    if (ctx->want_read_iters) { riters = ctx->ws_get<uint32_t>("riters", n); PS_HIP(hipMemsetAsync(riters, 0, (size_t)n * 4, s)); a.read_iters = riters; }
    { EvTimer t(s); launch_backtrack(a, ctx->ws_get<BtArgs>("btargs", 1), blocks, lm, s); PS_HIP(hipGetLastError()); b.tm.ms_backtrack += t.stop(); ++b.tm.n_backtrack_launches; }
    // compact the hit lists on the device, then one download through pinned staging
    EvTimer tc(s);
    uint32_t *cnt = ctx->ws_get<uint32_t>("cnt", n), *off = ctx->ws_get<uint32_t>("off", (size_t)n + 1);
    hipLaunchKernelGGL(k_clip_counts, dim3((n + 255) / 256), dim3(256), 0, s, n_aln, aln_cap, n, cnt);
    size_t tb = 0;
    PS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt, off, n, s));
    uint8_t *tmp = ctx->ws_get<uint8_t>("scan_tmp", tb ? tb : 1);
    PS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cnt, off, n, s));
    if (ctx->want_read_iters) {
        ctx->read_iters.resize(n); PS_HIP(hipMemcpyAsync(ctx->read_iters.data(), riters, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    }
    int32_t *p_na = (int32_t *)out.pn->get((size_t)n * 4 + 64); uint8_t *p_st = (uint8_t *)out.ps->get((size_t)n + 64);
    uint32_t *p_off = (uint32_t *)out.po->get(((size_t)n + 1) * 4 + 64);
    PS_HIP(hipMemcpyAsync(p_na, n_aln, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    PS_HIP(hipMemcpyAsync(p_st, status, (size_t)n, hipMemcpyDeviceToHost, s));
    PS_HIP(hipMemcpyAsync(p_off, off, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    PS_HIP(hipStreamSynchronize(s));
    uint32_t last = 0;
    if (n) { int m = p_na[n - 1]; last = p_off[n - 1] + (uint32_t)(m > aln_cap ? aln_cap : (m < 0 ? 0 : m)); }
    p_off[n] = last;
    AlnRec *p_al = (AlnRec *)out.pa->get((size_t)last * sizeof(AlnRec) + 64);
    if (last) {
        AlnRec *comp = ctx->ws_get<AlnRec>("comp", last);
        hipLaunchKernelGGL(k_gather_alns, dim3((n + 255) / 256), dim3(256), 0, s, alns, aln_cap, n_aln, off, n, comp);
        PS_HIP(hipMemcpyAsync(p_al, comp, (size_t)last * sizeof(AlnRec), hipMemcpyDeviceToHost, s));
        PS_HIP(hipStreamSynchronize(s));
    }
    out.n_aln = p_na; out.status = p_st; out.off = p_off; out.alns = p_al;
    b.tm.ms_compact += tc.stop();
}

void batch_search(Batch &b)
{
    Ctx *ctx = b.ctx;
    require_device(ctx->device);
    b.tm = Timing();
    auto t0 = Clock::now();
    b.d_stats.zero(ctx->stream);
    for (int t = 0; t < 3; ++t) b.n_overflow[t] = 0;
    for (Bin &bin : b.bins) {
        const int n = (int)bin.ids.size();
        if (!bin.pin_n_aln) { bin.pin_n_aln.reset(new PinBuf()); bin.pin_status.reset(new PinBuf()); bin.pin_off.reset(new PinBuf()); bin.pin_alns.reset(new PinBuf()); }
        SearchOut so{bin.pin_n_aln.get(), bin.pin_status.get(), bin.pin_off.get(), bin.pin_alns.get(), nullptr, nullptr, nullptr, nullptr};
        run_search(b, bin.md, n, bin.bases.p, bin.nmask.p, ctx->pool_cap[0], ctx->aln_cap[0], so);
        bin.h_n_aln = so.n_aln; bin.h_off = so.off; bin.h_alns = so.alns;
        bin.overflow.clear();
        std::vector<int32_t> todo;
        for (int r = 0; r < n; ++r) {
            if (so.status[r] == RS_OK) continue;
            if (so.status[r] == RS_BAD_SCORE) throw Error("internal: score outside the bucket range");
            todo.push_back(r);
        }
        for (int tier = 1; tier < 3 && !todo.empty(); ++tier) {
            b.n_overflow[tier] += (int64_t)todo.size();
            const int m = (int)todo.size();
            std::vector<uint32_t> hb((size_t)bin.n_bw * m), hm((size_t)bin.n_mw * m);
            for (int q = 0; q < m; ++q) {
                for (int wv = 0; wv < bin.n_bw; ++wv) hb[(size_t)wv * m + q] = bin.h_bases[(size_t)wv * n + todo[q]];
                for (int wv = 0; wv < bin.n_mw; ++wv) hm[(size_t)wv * m + q] = bin.h_nmask[(size_t)wv * n + todo[q]];
            }
            DevBuf<uint32_t> db, dm; db.alloc(hb.size()); dm.alloc(hm.size());
            db.upload(hb.data(), hb.size(), ctx->stream); dm.upload(hm.data(), hm.size(), ctx->stream);
            PinBuf t_n, t_s, t_o, t_a;
            SearchOut s2{&t_n, &t_s, &t_o, &t_a, nullptr, nullptr, nullptr, nullptr};
            run_search(b, bin.md, m, db.p, dm.p, ctx->pool_cap[tier], ctx->aln_cap[tier], s2);
            std::vector<int32_t> still;
            for (int q = 0; q < m; ++q) {
                if (s2.status[q] == RS_BAD_SCORE) throw Error("internal: score outside the bucket range");
                if (s2.status[q] != RS_OK) { still.push_back(todo[q]); continue; }
                bin.h_n_aln[todo[q]] = s2.n_aln[q];
                bin.overflow[todo[q]] = std::vector<AlnRec>(s2.alns + s2.off[q], s2.alns + s2.off[q + 1]);
            }
            todo.swap(still);
        }
        if (!todo.empty()) throw Error("a read exceeded the largest search tier (stack or hit capacity)");
    }
    KStats hs[3];
    b.d_stats.download(hs, 3, ctx->stream);
    PS_HIP(hipStreamSynchronize(ctx->stream));
    b.st_width = hs[0]; b.st_backtrack = hs[1];
    // classify reads for the tie-break stream: a read whose best score is reached by exactly one SA
    // interval always consumes two draws; the others ("hard") are data dependent
    auto tcl = Clock::now();
    const int64_t N = b.rs.n;
    b.n_best.assign((size_t)N, 0); b.hard.clear(); b.easy_before.clear(); b.n_easy = 0;
    for (int64_t g = 0; g < N; ++g) {
        int na; const AlnRec *al = b.alns_of(g, na);
        int nb = 0;
        for (; nb < na && al[nb].score == al[0].score; ++nb) {}
        b.n_best[g] = (uint8_t)(nb > 255 ? 255 : nb);
        if (nb == 1) ++b.n_easy;
        else if (nb >= 2) { b.hard.push_back(g); b.easy_before.push_back(b.n_easy); }
    }
    b.hits.resize((size_t)N);        // every record is rewritten by the selection stage
    b.tm.ms_classify = ms_since(tcl);
    b.searched = true; b.selected_hard = b.selected = b.located = false;
    b.tm.ms_total = ms_since(t0);
}

const AlnRec *Batch::alns_of(int64_t g, int &n) const
{
    const Bin &bin = bins[read_bin[g]];
    int32_t r = read_local[g];
    n = bin.h_n_aln[r];
    if (!bin.overflow.empty()) {
        auto it = bin.overflow.find(r);
        if (it != bin.overflow.end()) return it->second.data();
    }
    return bin.h_alns + bin.h_off[r];
}

// ----------------------------------------------------- tie-break selection -----
// Among the hits with the best score one occurrence is chosen at random; the reference's aligner
// draws from ONE drand48 stream (seed 11) over all reads in input order.
static int choose_main(const AlnRec *al, int na, Rng48 &rng, Hit &h)
{
    int cnt = 0, draws = 0, i;
    const int best = al[0].score;
    for (i = 0; i < na; ++i) {
        const AlnRec &p = al[i];
        if (p.score > best) break;
        const uint64_t wdt = (uint64_t)(p.l - p.k) + 1ull;
        ++draws;
        if (rng.drand() * (double)(wdt + (uint64_t)(int64_t)cnt) > (double)cnt) {
            h.n_mm = p.n_mm; h.n_gapo = p.n_gapo; h.n_gape = p.n_gape;
            h.ref_shift = (int)p.n_del - (int)p.n_ins; h.score = p.score;
            h.sa = p.k + (bwtint)((double)wdt * rng.drand());
            ++draws;
        }
        cnt += (int)wdt;
    }
    h.c1 = cnt;
    for (; i < na; ++i) cnt += (int)((uint64_t)(al[i].l - al[i].k) + 1ull);
    h.c2 = cnt - h.c1;
    h.type = h.c1 > 1 ? 2 : 1;
    return draws;
}

void batch_select_hard(Batch &b, uint64_t draws_before, uint64_t *draws_after)
{
    if (!b.searched) throw Error("select before search");
    auto t0 = Clock::now();
    b.draws_in = draws_before;
    b.hard_draws_cum.assign(b.hard.size(), 0);
    Rng48 rng(11);
    rng.jump(draws_before);
    uint64_t H = 0; int64_t e_prev = 0;
    for (size_t q = 0; q < b.hard.size(); ++q) {
        const int64_t g = b.hard[q];
        rng.jump(2ull * (uint64_t)(b.easy_before[q] - e_prev));   // the single-best reads in between took two draws each
        e_prev = b.easy_before[q];
        int na; const AlnRec *al = b.alns_of(g, na);
        Hit &h = b.hits[g];
        h = Hit();
        H += (uint64_t)choose_main(al, na, rng, h);
        b.hard_draws_cum[q] = H;
    }
    b.draws_out = draws_before + 2ull * (uint64_t)b.n_easy + H;
    if (draws_after) *draws_after = b.draws_out;
    b.selected_hard = true;
    b.tm.ms_select += ms_since(t0); b.tm.ms_sel_hard = ms_since(t0);
}

void batch_select_easy(Batch &b, int threads)
{
    if (!b.selected_hard) throw Error("select_easy before select_hard");
    auto t0 = Clock::now();
    const int64_t N = b.rs.n;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    const int n_occ = b.ctx->opt.n_occ;
    std::vector<std::vector<Multi>> mul_chunks((size_t)threads);
    std::vector<std::string> errs((size_t)threads);
    auto work = [&](int t) {
        const int64_t g0 = N * t / threads, g1 = N * (t + 1) / threads;
        // stream position at g0: single-best reads before it and hard reads before it
        size_t q = std::lower_bound(b.hard.begin(), b.hard.end(), g0) - b.hard.begin();
        int64_t easy = 0;
        for (int64_t g = 0; g < g0; ++g) easy += b.n_best[g] == 1;   // prefix count (cheap byte scan)
        Rng48 rng(11);
        rng.jump(b.draws_in + 2ull * (uint64_t)easy + (q ? b.hard_draws_cum[q - 1] : 0ull));
        std::vector<Multi> &mul = mul_chunks[t];
        for (int64_t g = g0; g < g1; ++g) {
            Hit &h = b.hits[g];
            int na; const AlnRec *al = b.alns_of(g, na);
            if (b.n_best[g] == 0) { h = Hit(); h.type = 0; h.pos = -1; h.multi_begin = (int32_t)mul.size(); continue; }
            if (b.n_best[g] == 1) {
                h = Hit();
                Rng48 probe = rng;
                if (probe.step() == 0) { errs[t] = "tie-break stream hit the zero state; sequential replay required"; return; }
                choose_main(al, na, rng, h);
            } else {
                const uint64_t prev = q ? b.hard_draws_cum[q - 1] : 0ull;
                const uint64_t used = b.hard_draws_cum[q] - prev;
                for (uint64_t d = 0; d < used; ++d) rng.step();   // selection already done in select_hard
                ++q;
            }
            h.pos = -1; h.multi_begin = (int32_t)mul.size(); h.n_multi = 0;
            // alternative hits (samse -n): listed only if all occurrences of all hits number <= n_occ+1
            if (n_occ > 0) {
                int tot = 0;
                for (int k = 0; k < na; ++k) tot += (int)((uint64_t)(al[k].l - al[k].k) + 1ull);
                if (tot >= 0 && tot <= n_occ + 1) {
                    for (int k = 0; k < na; ++k)
                        for (uint64_t row = al[k].k; row <= al[k].l; ++row) {
                            Multi m; std::memset(&m, 0, sizeof m);
                            m.row = (bwtint)row; m.gap = al[k].n_gapo + al[k].n_gape; m.mm = al[k].n_mm;
                            m.ref_shift = (int)al[k].n_del - (int)al[k].n_ins; m.pos = -1;
                            mul.push_back(m); ++h.n_multi;
                        }
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (auto &e : errs) if (!e.empty()) throw Error(e);
    // stitch the per-thread alternative-hit lists
    b.multis.clear();
    for (int t = 0; t < threads; ++t) {
        const int64_t g0 = N * t / threads, g1 = N * (t + 1) / threads;
        const int32_t base = (int32_t)b.multis.size();
        if (base) for (int64_t g = g0; g < g1; ++g) b.hits[g].multi_begin += base;
        b.multis.insert(b.multis.end(), mul_chunks[t].begin(), mul_chunks[t].end());
    }
    b.selected = true;
    b.tm.ms_select += ms_since(t0); b.tm.ms_sel_easy = ms_since(t0);
}

// ------------------------------------------------- locate / MAPQ / gapped DP ----
static int mapq_logn(int n) { return (int)(4.343 * std::log((double)n) + 0.5); }

static int approx_mapq(const Hit &h, const Options &o, int len)
{
    const int budget = budget_diffs(o, len);
    if (h.c1 == 0) return 23;
    if (h.c1 > 1) return 0;
    if (!o.profile) { if (h.n_mm == budget) return 25; }
    else if (budget * o.unit - h.score < o.unit) return 25;
    if (h.c2 == 0) return 37;
    const int n = h.c2 >= 255 ? 255 : h.c2;
    return 23 < mapq_logn(n) ? 0 : 23 - mapq_logn(n);
}

// text position of an SA row -> forward coordinate of the alignment's first base and its strand
static int64_t to_forward(uint32_t pos_f32, int64_t l_pac, int ref_len, int &strand)
{
    int64_t pos_f = (int64_t)pos_f32;
    strand = 0;
    if (pos_f < l_pac && l_pac < pos_f + ref_len) return -1;       // spans the forward/reverse junction
    const bool is_rev = pos_f >= l_pac;
    if (is_rev) pos_f = 2 * l_pac - 1 - pos_f;
    strand = !is_rev;
    if (is_rev) pos_f = pos_f + 1 < ref_len ? 0 : pos_f - ref_len + 1;
    return pos_f;
}

static int fix_cigar(uint32_t *cigar, int n, int64_t &rb)
{
    if (n <= 0) return 0;
    if ((cigar[n - 1] & 0xf) == 1) cigar[n - 1] = (cigar[n - 1] >> 4 << 4) | 3;   // trailing insertion -> soft clip
    if ((cigar[0] & 0xf) == 1) cigar[0] = (cigar[0] >> 4 << 4) | 3;
    if ((cigar[n - 1] & 0xf) == 2) --n;                                            // trailing deletion dropped
    if (n > 0 && (cigar[0] & 0xf) == 2) { rb += cigar[0] >> 4; --n; std::memmove(cigar, cigar + 1, (size_t)n * 4); }
    return n;
}

void batch_locate(Batch &b)
{
    if (!b.selected) throw Error("locate before select");
    Ctx *ctx = b.ctx; hipStream_t s = ctx->stream;
    require_device(ctx->device);
    const int64_t N = b.rs.n, l_pac = ctx->ix.ref.l_pac;
    auto t0 = Clock::now();
    // rows to locate: one slot per read (row 0 = nothing to walk for unmapped reads), then the alternatives
    const size_t n_rows = (size_t)N + b.multis.size(), multi_base = (size_t)N;
    bwtint *p_rows = ctx->pin_get<bwtint>("rows", n_rows + 1);
    {
        int nt = std::max(1, std::min(ctx->host_threads, 64));
        auto fill = [&](int t) {
            for (int64_t g = N * t / nt; g < N * (t + 1) / nt; ++g) p_rows[g] = b.hits[g].type != 0 ? b.hits[g].sa : 0;
            const int64_t M = (int64_t)b.multis.size();
            for (int64_t j = M * t / nt; j < M * (t + 1) / nt; ++j) p_rows[multi_base + j] = b.multis[j].row;
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(fill, t);
        fill(0);
        for (auto &x : th) x.join();
    }
    b.tm.ms_rows = ms_since(t0);
    bwtint *pos = ctx->pin_get<bwtint>("pos", n_rows + 1);
    if (n_rows) {
        bwtint *d_rows = ctx->ws_get<bwtint>("rows", n_rows), *d_pos = ctx->ws_get<bwtint>("pos", n_rows);
        PS_HIP(hipMemcpyAsync(d_rows, p_rows, n_rows * sizeof(bwtint), hipMemcpyHostToDevice, s));
        { EvTimer t(s); launch_sa2pos(ctx->ix.view, d_rows, d_pos, (int)n_rows, b.d_stats.p + 2, s); PS_HIP(hipGetLastError()); b.tm.ms_sa2pos += t.stop(); }
        PS_HIP(hipMemcpyAsync(pos, d_pos, n_rows * sizeof(bwtint), hipMemcpyDeviceToHost, s));
        PS_HIP(hipStreamSynchronize(s));
        PS_HIP(hipMemcpy(&b.st_sa2pos, b.d_stats.p + 2, sizeof(KStats), hipMemcpyDeviceToHost));
    }
    auto t1 = Clock::now();
    std::vector<std::vector<RefineItem>> items(b.bins.size());
    struct Back { int64_t g; int32_t multi; };             // multi < 0: main hit
    std::vector<std::vector<Back>> back(b.bins.size());
    {   // strand / MAPQ / alternative-hit filter: independent per read
        int nt = std::max(1, std::min(ctx->host_threads, 64));
        auto work = [&](int t) {
            for (int64_t g = N * t / nt; g < N * (t + 1) / nt; ++g) {
                Hit &h = b.hits[g];
                const int len = b.rs.len[g];
                if (h.type != 0) {
                    int strand = 0;
                    h.pos = to_forward(pos[g], l_pac, len + h.ref_shift, strand);
                    h.strand = strand;
                    h.mapq = approx_mapq(h, ctx->opt, len);
                    if (h.pos < 0) h.type = 0;
                }
                int kept = 0;
                for (int j = 0; j < h.n_multi; ++j) {
                    Multi &m = b.multis[h.multi_begin + j];
                    int strand = 0;
                    m.pos = to_forward(pos[multi_base + h.multi_begin + j], l_pac, len + m.ref_shift, strand);
                    m.strand = strand;
                    if (m.pos != h.pos && m.pos >= 0) b.multis[h.multi_begin + kept++] = m;
                }
                h.n_multi = kept;
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (int64_t g = 0; g < N; ++g) {                       // gapped hits go to the banded-DP kernel
        Hit &h = b.hits[g];
        if (h.n_multi == 0 && !(h.type != 0 && h.n_gapo)) continue;
        const int bi = b.read_bin[g];
        for (int j = 0; j < h.n_multi; ++j) {
            Multi &m = b.multis[h.multi_begin + j];
            if (m.gap) { items[bi].push_back(RefineItem{b.read_local[g], (bwtint)m.pos, m.ref_shift, m.strand}); back[bi].push_back(Back{g, j}); }
        }
        if (h.type != 0 && h.n_gapo) { items[bi].push_back(RefineItem{b.read_local[g], (bwtint)h.pos, h.ref_shift, h.strand}); back[bi].push_back(Back{g, -1}); }
    }
    b.tm.ms_host_post += ms_since(t1);
    // banded DP kernel, one launch per length bin
    for (size_t bi = 0; bi < b.bins.size(); ++bi) {
        const int n_it = (int)items[bi].size();
        if (!n_it) continue;
        Bin &bin = b.bins[bi];
        DevBuf<RefineItem> d_it; DevBuf<uint32_t> d_cig; DevBuf<int32_t> d_nc; DevBuf<uint8_t> zbuf;
        d_it.alloc(n_it); d_cig.alloc((size_t)n_it * PS_MAX_CIGAR); d_nc.alloc(n_it);
        d_it.upload(items[bi].data(), n_it, s);
        int blocks = (n_it + 63) / 64; if (blocks > 2048) blocks = 2048;
        const int tmax = bin.len + 64;
        RefineArgs ra;
        ra.ix = ctx->ix.view; ra.n_items = n_it; ra.len = bin.len; ra.n_reads = (int)bin.ids.size();
        ra.bases = bin.bases.p; ra.nmask = bin.nmask.p; ra.items = d_it.p; ra.cigar = d_cig.p; ra.n_cigar = d_nc.p;
        ra.z_per_block = (size_t)64 * tmax * (bin.len < 2 * tmax + 1 ? bin.len : 2 * tmax + 1);
        zbuf.alloc(ra.z_per_block * blocks); ra.zbuf = zbuf.p;
        { EvTimer t(s); launch_refine(ra, blocks, s); PS_HIP(hipGetLastError()); b.tm.ms_refine += t.stop(); }
        std::vector<uint32_t> cig((size_t)n_it * PS_MAX_CIGAR); std::vector<int32_t> nc(n_it);
        d_cig.download(cig.data(), cig.size(), s); d_nc.download(nc.data(), n_it, s);
        PS_HIP(hipStreamSynchronize(s));
        for (int q = 0; q < n_it; ++q) {
            const Back &bk = back[bi][q];
            Hit &h = b.hits[bk.g];
            uint32_t *c = cig.data() + (size_t)q * PS_MAX_CIGAR;
            if (bk.multi < 0) {
                int64_t rb = h.pos;
                h.n_cigar = fix_cigar(c, nc[q], rb);
                if (h.n_cigar > PS_HIT_CIGAR) throw Error("CIGAR with more than 8 operations (raise PS_HIT_CIGAR for max_gapo > 2)");
                std::memcpy(h.cigar, c, sizeof h.cigar);
                h.pos = rb;
                if (h.n_cigar == 0) h.type = 0;
            } else {
                Multi &m = b.multis[h.multi_begin + bk.multi];
                int64_t rb = m.pos;
                m.n_cigar = fix_cigar(c, nc[q], rb);
                std::memcpy(m.cigar, c, sizeof m.cigar);
                m.pos = rb;
            }
        }
    }
    // alternatives whose gapped refinement produced nothing are dropped
    for (int64_t g = 0; g < N; ++g) {
        Hit &h = b.hits[g];
        int kept = 0;
        for (int j = 0; j < h.n_multi; ++j) {
            Multi &m = b.multis[h.multi_begin + j];
            if (m.gap && m.n_cigar == 0) continue;
            b.multis[h.multi_begin + kept++] = m;
        }
        h.n_multi = kept;
    }
    b.located = true;
    b.tm.ms_total += ms_since(t0);
}

// ------------------------------------------------------------------ SAM -------
static inline int host_pac(const std::vector<uint8_t> &pac, int64_t p) { return (pac[(size_t)p >> 2] >> ((~p & 3) << 1)) & 3; }
static void put_int(std::string &o, long v) { char b[24]; int l = std::snprintf(b, sizeof b, "%ld", v); o.append(b, (size_t)l); }
static void put_cigar(std::string &o, int n, const uint32_t *c, int len)
{
    if (n) for (int j = 0; j < n; ++j) { put_int(o, c[j] >> 4); o.push_back("MIDS"[c[j] & 0xf]); }
    else { put_int(o, len); o.push_back('M'); }
}
static int64_t ref_span(int n, const uint32_t *c, int len)
{
    if (!n) return len;
    int64_t x = 0;
    for (int j = 0; j < n; ++j) { int op = c[j] & 0xf; if (op == 0 || op == 2) x += c[j] >> 4; }
    return x;
}
// MD string and edit distance by direct comparison with the reference
static void cal_md(const RefSeq &ref, int n_cigar, const uint32_t *cigar, int len, int64_t pos, const uint8_t *seq, std::string &md, int &nm)
{
    int64_t x = pos, y = 0; int u = 0; nm = 0; md.clear();
    auto cmp = [&](int l) {
        for (int z = 0; z < l && x + z < ref.l_pac; ++z) {
            int c = host_pac(ref.pac, x + z);
            if (seq[y + z] > 3 || c != seq[y + z]) { put_int(md, u); md.push_back("ACGTN"[c]); ++nm; u = 0; } else ++u;
        }
    };
    if (n_cigar) {
        for (int k = 0; k < n_cigar; ++k) {
            int l = (int)(cigar[k] >> 4), op = (int)(cigar[k] & 0xf);
            if (op == 0) { cmp(l); x += l; y += l; }
            else if (op == 1 || op == 3) { y += l; if (op == 1) nm += l; }
            else if (op == 2) {
                put_int(md, u); md.push_back('^');
                for (int z = 0; z < l && x + z < ref.l_pac; ++z) md.push_back("ACGT"[host_pac(ref.pac, x + z)]);
                u = 0; x += l; nm += l;
            }
        }
    } else cmp(len);
    put_int(md, u);
}

static void sam_line(const Batch &b, int64_t g, std::string &o)
{
    const ReadSet &rs = b.rs; const RefSeq &ref = b.ctx->ix.ref; const Options &opt = b.ctx->opt;
    const Hit &h = b.hits[g];
    const int len = rs.len[g];
    const uint8_t *seq = rs.seq.data() + rs.off[g];
    const char *qual = rs.has_qual ? rs.qual.data() + rs.off[g] : nullptr;
    size_t nl; const char *nm_ = rs.name(g, nl);
    o.append(nm_, nl);
    auto put_seq = [&](int strand) {
        if (!strand) for (int i = 0; i < len; ++i) o.push_back("ACGTN"[seq[i]]);
        else for (int i = len - 1; i >= 0; --i) o.push_back("TGCAN"[seq[i]]);
        o.push_back('\t');
        if (qual) { if (!strand) o.append(qual, (size_t)len); else for (int i = len - 1; i >= 0; --i) o.push_back(qual[i]); }
        else o.push_back('*');
    };
    if (h.type == 0) { o.append("\t4\t*\t0\t0\t*\t*\t0\t0\t"); put_seq(h.strand); o.push_back('\n'); return; }
    int seqid = 0, flag = 0;
    const int span = (int)ref_span(h.n_cigar, h.cigar, len);
    const int nn = ref.cnt_ambi(h.pos, span, &seqid);
    const Contig &ct = ref.contigs[seqid];
    if (h.pos + span - ct.offset > ct.len) flag |= 4;      // bridges two reference sequences
    if (h.strand) flag |= 16;
    o.push_back('\t'); put_int(o, flag); o.push_back('\t'); o.append(ct.name); o.push_back('\t');
    put_int(o, (long)(h.pos - ct.offset + 1)); o.push_back('\t'); put_int(o, h.mapq); o.push_back('\t');
    put_cigar(o, h.n_cigar, h.cigar, len);
    o.append("\t*\t0\t0\t");
    put_seq(h.strand);
    // oriented read for MD/NM
    std::vector<uint8_t> tmp;
    const uint8_t *oriented = seq;
    if (h.strand) { tmp.resize((size_t)len); for (int i = 0; i < len; ++i) { uint8_t c = seq[len - 1 - i]; tmp[i] = c > 3 ? c : (uint8_t)(3 - c); } oriented = tmp.data(); }
    std::string md; int nm = 0;
    cal_md(ref, h.n_cigar, h.cigar, len, h.pos, oriented, md, nm);
    char XT = "NURM"[h.type];
    if (nn > 10) XT = 'N';
    o.append("\tXT:A:"); o.push_back(XT); o.append("\tNM:i:"); put_int(o, nm);
    if (nn) { o.append("\tXN:i:"); put_int(o, nn); }
    o.append("\tX0:i:"); put_int(o, h.c1);
    if (h.c1 <= opt.max_top2) { o.append("\tX1:i:"); put_int(o, h.c2); }
    o.append("\tXM:i:"); put_int(o, h.n_mm); o.append("\tXO:i:"); put_int(o, h.n_gapo); o.append("\tXG:i:"); put_int(o, h.n_gapo + h.n_gape);
    o.append("\tMD:Z:"); o.append(md);
    if (h.n_multi) {
        o.append("\tXA:Z:");
        for (int j = 0; j < h.n_multi; ++j) {
            const Multi &m = b.multis[h.multi_begin + j];
            int sid = 0;
            ref.cnt_ambi(m.pos, (int)ref_span(m.n_cigar, m.cigar, len), &sid);
            const Contig &mc = ref.contigs[sid];
            o.append(mc.name); o.push_back(','); o.push_back(m.strand ? '-' : '+'); put_int(o, (long)(m.pos - mc.offset + 1)); o.push_back(',');
            put_cigar(o, m.n_cigar, m.cigar, len);
            o.push_back(','); put_int(o, m.gap + m.mm); o.push_back(';');
        }
    }
    o.push_back('\n');
}

void batch_write_sam(Batch &b, const char *path, bool header, const char *pg_line, int threads)
{
    if (!b.located) throw Error("write_sam before locate");
    FILE *f = std::fopen(path, "wb");
    if (!f) throw Error(std::string("cannot write ") + path);
    if (header) {
        for (const Contig &c : b.ctx->ix.ref.contigs) std::fprintf(f, "@SQ\tSN:%s\tLN:%d\n", c.name.c_str(), c.len);
        if (pg_line && pg_line[0]) std::fprintf(f, "%s\n", pg_line);
    }
    const int64_t N = b.rs.n;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    const int64_t chunk = 1 << 16;
    for (int64_t base = 0; base < N; base += chunk * threads) {
        std::vector<std::string> out((size_t)threads);
        auto work = [&](int t) {
            int64_t g0 = base + chunk * t, g1 = std::min(N, g0 + chunk);
            std::string &o = out[t];
            if (g0 < g1) o.reserve((size_t)(g1 - g0) * 256);
            for (int64_t g = g0; g < g1; ++g) sam_line(b, g, o);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        for (auto &o : out) if (!o.empty() && std::fwrite(o.data(), 1, o.size(), f) != o.size()) { std::fclose(f); throw Error(std::string("short write on ") + path); }
    }
    if (std::fclose(f) != 0) throw Error(std::string("cannot close ") + path);
}

}  // namespace ps
