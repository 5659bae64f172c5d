// ps_capi.hip -- extern "C" boundary (include/parasuite_hip.h).  Exceptions stop here.
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <fcntl.h>
#include <unistd.h>
#include "../../include/parasuite_hip.h"
#include "ps_pipeline.h"
#include "ps_bam.h"

using namespace ps;

static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; std::fprintf(stderr, "[parasuite-hip] error: %s\n", m.c_str()); return 1; }
#define PS_TRY try {
#define PS_CATCH_INT } catch (const std::exception &e) { return fail(e.what()); } catch (...) { return fail("unknown error"); }
#define PS_CATCH_PTR } catch (const std::exception &e) { fail(e.what()); return nullptr; } catch (...) { fail("unknown error"); return nullptr; }

struct ps_ctx { Ctx c; };
struct ps_batch { std::unique_ptr<Batch> b; };

static_assert(sizeof(ps_aln) == sizeof(AlnRec), "ps_aln layout");

// Written pieces of ps_map are freed by threads of their own (gigabytes of host memory per piece: 0.18 s for 7.7 M reads), and the call
// does not wait for the last of them: they are joined by the next call, by ps_release_host_cache and when the process exits.
namespace {
struct Trash {
    std::mutex mu; std::vector<std::thread> th; bool hooked = false;
    void add(std::thread &&t) { std::lock_guard<std::mutex> l(mu); th.push_back(std::move(t)); if (!hooked) { hooked = true; std::atexit([]() { trash().collect(); }); } }
    void collect() { std::vector<std::thread> all; { std::lock_guard<std::mutex> l(mu); all.swap(th); } for (auto &t : all) if (t.joinable()) t.join(); }
    static Trash &trash() { static Trash *t = new Trash(); return *t; }       // never destroyed: a thread may still run at exit
};
}
// bounded hand-over between the stages of ps_map
namespace {
template <class T> struct Chan {
    std::mutex m; std::condition_variable cv; std::deque<T> q; bool closed = false; size_t cap = 2; int waiting = 0;
    void push(T &&v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return q.size() < cap || closed; }); if (closed) return; q.push_back(std::move(v)); cv.notify_all(); }
    bool pop(T &v) { std::unique_lock<std::mutex> l(m); ++waiting; cv.wait(l, [&] { return !q.empty() || closed; }); --waiting; if (q.empty()) return false; v = std::move(q.front()); q.pop_front(); cv.notify_all(); return true; }
    bool hungry() { std::lock_guard<std::mutex> l(m); return q.empty() && waiting > 0; }     // somebody waits for work and there is none
    void close() { std::lock_guard<std::mutex> l(m); closed = true; cv.notify_all(); }      // what is queued is still handed out
    void abort() { std::lock_guard<std::mutex> l(m); closed = true; q.clear(); cv.notify_all(); }
};
}

extern "C" {

const char *ps_version(void) { return "parasuite-hip 0.1 (gfx950)"; }
const char *ps_last_error(void) { return g_err.c_str(); }

// A context in two steps: options and knobs (no device call: ps_map's parser needs nothing else and starts before the runtime is
// up, which takes 0.2-0.3 s in a fresh process), then the device side (stream, clock).
static void ctx_attach_device(ps_ctx *x)
{
    require_device(x->c.device);
    if (x->c.stream) return;
    PS_HIP(hipStreamCreateWithFlags(&x->c.stream, hipStreamNonBlocking));
    PS_HIP(hipEventCreate(&x->c.ref_event)); PS_HIP(hipEventRecord(x->c.ref_event, x->c.stream)); PS_HIP(hipEventSynchronize(x->c.ref_event));
}
static ps_ctx *new_ctx(int device, bool attach = true)
{
    ps_ctx *x = new ps_ctx();
    x->c.device = device;
    if (attach) { try { ctx_attach_device(x); } catch (...) { delete x; throw; } }
    if (const char *e = std::getenv("PS_FETCH_MIN")) x->c.fetch_min = std::atoi(e);       // tuning knobs
    if (const char *e = std::getenv("PS_N_BIG")) x->c.n_big = std::atoi(e);
    if (const char *e = std::getenv("PS_HIT_MIN")) x->c.hit_min = std::atoi(e);
    if (std::getenv("PS_READ_ITERS")) x->c.want_read_iters = true;
    if (std::getenv("PS_KSTATS")) x->c.want_kstats = true;
    if (const char *e = std::getenv("PS_BT_BLOCKS")) x->c.bt_blocks = std::atoi(e);
    if (const char *e = std::getenv("PS_POOL_CAP")) x->c.pool_cap[0] = (uint32_t)std::atoi(e);
    if (const char *e = std::getenv("PS_ALN_CAP")) { x->c.aln_cap[0] = std::max(1, std::atoi(e)); x->c.aln_cap_short = 0; }   // hit intervals a read may list in the first tier (stated: for every length)
    return x;
}

// a second context on `device` holding a COPY of src's index (blobs + jump table, device to device: the route ps_map takes for every
// device after the first; src and the new context may be on the same device)
ps_ctx *ps_ctx_clone(ps_ctx *src, int device)
{
    ps_ctx *x = nullptr;
    PS_TRY
        x = new_ctx(device);
        index_clone(src->c.ix, src->c.device, x->c.ix, device, x->c.stream);
        return x;
    } catch (const std::exception &e) { delete x; fail(e.what()); return nullptr; } catch (...) { delete x; fail("unknown error"); return nullptr; }
}

ps_ctx *ps_ctx_open(const char *ref_fa, int device)
{
    ps_ctx *x = nullptr;
    PS_TRY
        x = new_ctx(device);
        index_load(ref_fa, x->c.ix, x->c.stream);
        return x;
    } catch (const std::exception &e) { delete x; fail(e.what()); return nullptr; } catch (...) { delete x; fail("unknown error"); return nullptr; }
}
ps_ctx *ps_ctx_build(const char *ref_fa, int device, int save_files)
{
    ps_ctx *x = nullptr;
    PS_TRY
        x = new_ctx(device);
        index_build(ref_fa, x->c.ix, x->c.stream);
        if (save_files) index_save(x->c.ix, ref_fa);
        return x;
    } catch (const std::exception &e) { delete x; fail(e.what()); return nullptr; } catch (...) { delete x; fail("unknown error"); return nullptr; }
}
void ps_ctx_close(ps_ctx *x) { delete x; }

int ps_index(const char *ref_fa)
{
    PS_TRY
        ps_ctx *x = ps_ctx_build(ref_fa, 0, 1);
        if (!x) return 1;
        delete x;
        return 0;
    PS_CATCH_INT
}

int ps_ctx_set_stock(ps_ctx *x, const char *n_arg)
{
    PS_TRY
        Options o; set_stock_n(o, n_arg);
        if (o.max_diff < 0 && !(o.fnr > 0.0)) throw Error("bad -n argument");
        x->c.opt = o; return 0;
    PS_CATCH_INT
}
int ps_ctx_set_profile_matrix(ps_ctx *x, const double P[16], double ins, double del, int xarg)
{
    PS_TRY
        Options o; profile_costs(o, P, ins, del, xarg); x->c.opt = o; return 0;
    PS_CATCH_INT
}
int ps_ctx_set_profile(ps_ctx *x, const char *ep, const char *ip, const char *x_arg)
{
    PS_TRY
        double P[16], ins, del; std::string err;
        if (!read_profile_files(ep, ip, P, ins, del, err)) throw Error(err);
        return ps_ctx_set_profile_matrix(x, P, ins, del, x_arg ? std::atoi(x_arg) : -1);
    PS_CATCH_INT
}
// SA[row] for arbitrary rows of the BW matrix (LF walk to a sampled row: the kernel the samse stage uses): index checks
// every row of the index against the text (ps_kernels.hip: k_index_check): out = rows visited (must be seq_len + 1), BWT symbols that differ
// from the text, SA samples that differ from the position counted along the LF cycle, longest arc between two samples
int ps_ctx_index_check(ps_ctx *x, uint64_t out[4])
{
    PS_TRY
        Ctx &c = x->c;
        require_device(c.device);
        DevBuf<unsigned long long> d; d.alloc(4);
        PS_HIP(hipMemsetAsync(d.p, 0, 32, c.stream));
        launch_index_check(c.ix.view, d.p, c.stream);
        PS_HIP(hipGetLastError());
        unsigned long long h[4];
        d.download(h, 4, c.stream);
        PS_HIP(hipStreamSynchronize(c.stream));
        for (int j = 0; j < 4; ++j) out[j] = h[j];
        return 0;
    PS_CATCH_INT
}
int ps_ctx_sa_lookup(ps_ctx *x, const uint64_t *rows, int64_t n, uint64_t *out)
{
    PS_TRY
        Ctx &c = x->c;
        require_device(c.device);
        if (n < 0 || n > 0x7fffffff) throw Error("sa lookup: bad count");
        for (int64_t i = 0; i < n; ++i) if (rows[i] < 1 || rows[i] > c.ix.view.seq_len) throw Error("sa lookup: row outside [1, n]");
        DevBuf<bwtint> d_r, d_p; d_r.alloc((size_t)std::max<int64_t>(1, n)); d_p.alloc((size_t)std::max<int64_t>(1, n));
        if (n) {
            d_r.upload((const bwtint *)rows, (size_t)n, c.stream);
            launch_sa2pos(c.ix.view, d_r.p, d_p.p, (int)n, nullptr, c.stream);
            PS_HIP(hipGetLastError());
            d_p.download((bwtint *)out, (size_t)n, c.stream);
        }
        PS_HIP(hipStreamSynchronize(c.stream));
        return 0;
    PS_CATCH_INT
}
int ps_ctx_set_lanes(ps_ctx *x, int n)
{
    PS_TRY
        if (n < 1 || n > (int)Ctx::N_WORK) throw Error("lanes: 1 to 4");
        x->c.n_work = n; return 0;
    PS_CATCH_INT
}
int ps_ctx_set_stats(ps_ctx *x, int on)
{
    PS_TRY
        x->c.want_kstats = on != 0; return 0;
    PS_CATCH_INT
}
int ps_ctx_set_tiers(ps_ctx *x, const uint32_t pool_cap[3], const int32_t aln_cap[3], int bt_blocks)
{
    PS_TRY
        for (int t = 0; t < 3; ++t) { if (pool_cap) x->c.pool_cap[t] = pool_cap[t]; if (aln_cap) x->c.aln_cap[t] = aln_cap[t]; }
        if (aln_cap) x->c.aln_cap_short = (aln_cap[0] == 8) ? 32 : 0;                  // stated sizes hold for every length; the default gets its short-read rule back
        x->c.bt_blocks = bt_blocks; return 0;
    PS_CATCH_INT
}
int ps_ctx_info(ps_ctx *x, ps_index_info *o)
{
    PS_TRY
        const Index &ix = x->c.ix;
        o->seq_len = ix.view.seq_len; o->l_pac = ix.view.l_pac; o->primary = ix.view.primary;
        for (int j = 0; j < 5; ++j) o->L2[j] = ix.view.L2[j];
        o->n_blocks = ix.view.n_blocks; o->n_sa = ix.view.n_sa; o->device_bytes = ix.device_bytes();
        o->n_contigs = (int)ix.ref.contigs.size(); o->n_holes = (int)ix.ref.holes.size(); o->sa_rounds = ix.sa_rounds; o->sa_intv = ix.view.sa_intv;
        o->build_ms = ix.build_ms; o->jump_levels = ix.jump_levels; o->pad_ = 0; return 0;
    PS_CATCH_INT
}
int ps_ctx_blob(ps_ctx *x, int which, void **p, uint64_t *bytes)
{
    PS_TRY
        Index &ix = x->c.ix;
        if (which == 0) { *p = ix.blocks.p; *bytes = ix.blocks.n * sizeof(OccBlock); }
        else if (which == 1) { *p = ix.sa.p; *bytes = ix.sa.n * sizeof(uint32_t); }
        else if (which == 2) { *p = ix.pac.p; *bytes = ix.pac.n; }
        else throw Error("blob index out of range");
        return 0;
    PS_CATCH_INT
}
int64_t ps_ctx_meta(ps_ctx *x, char *buf, int64_t cap)
{
    try {
        std::string m = index_meta_serialize(x->c.ix);
        if (buf && cap >= (int64_t)m.size()) std::memcpy(buf, m.data(), m.size());
        return (int64_t)m.size();
    } catch (const std::exception &e) { fail(e.what()); return -1; }
}
ps_ctx *ps_ctx_from_blobs(const char *meta, int64_t meta_len, int device, void *const ptrs[3])
{
    ps_ctx *x = nullptr;
    PS_TRY
        x = new_ctx(device);
        Index &ix = x->c.ix;
        index_meta_deserialize(std::string(meta, (size_t)meta_len), ix);
        const bwtint primary = ix.view.primary; bwtint L2[5]; std::memcpy(L2, ix.view.L2, sizeof L2);
        const size_t nb = ix.view.n_blocks, ns = ix.view.n_sa, np = (size_t)ix.ref.l_pac / 4 + 1;
        ix.blocks.adopt((OccBlock *)ptrs[0], nb); ix.sa.adopt((uint32_t *)ptrs[1], Index::sa_words(ns)); ix.pac.adopt((uint8_t *)ptrs[2], np);
        ix.ref.pac.resize(np);
        PS_HIP(hipMemcpy(ix.ref.pac.data(), ix.pac.p, np, hipMemcpyDeviceToHost));
        ix.refresh_view();
        ix.view.primary = primary; std::memcpy(ix.view.L2, L2, sizeof L2);
        index_build_jump(ix, nullptr);
        return x;
    } catch (const std::exception &e) { delete x; fail(e.what()); return nullptr; } catch (...) { delete x; fail("unknown error"); return nullptr; }
}
int ps_ctx_fetch(ps_ctx *x, int which, void *dst, uint64_t bytes)
{
    PS_TRY
        void *p; uint64_t n;
        if (ps_ctx_blob(x, which, &p, &n)) return 1;
        if (bytes > n) throw Error("fetch larger than blob");
        PS_HIP(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
        return 0;
    PS_CATCH_INT
}

int ps_ctx_export_blob(ps_ctx *x, int which, void *dst, uint64_t bytes)
{
    PS_TRY
        void *p; uint64_t n;
        if (ps_ctx_blob(x, which, &p, &n)) return 1;
        if (bytes != n) throw Error("export size differs from blob size");
        PS_HIP(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToDevice));
        return 0;
    PS_CATCH_INT
}

ps_batch *ps_batch_from_fastq(ps_ctx *x, const char *fastq)
{
    PS_TRY
        require_device(x->c.device);
        ReadSet rs; load_reads(fastq, rs, x->c.host_threads);
        ps_batch *b = new ps_batch();
        b->b = batch_create(&x->c, std::move(rs));
        return b;
    PS_CATCH_PTR
}
ps_batch *ps_batch_from_codes(ps_ctx *x, int64_t n, int len, const uint8_t *codes)
{
    PS_TRY
        require_device(x->c.device);
        ReadSet rs; reads_from_codes(n, len, codes, rs);
        ps_batch *b = new ps_batch();
        b->b = batch_create(&x->c, std::move(rs));
        return b;
    PS_CATCH_PTR
}
void ps_batch_free(ps_batch *b) { delete b; }
int64_t ps_batch_n(ps_batch *b) { return b->b->rs.n; }
int ps_batch_search(ps_batch *b) { PS_TRY batch_search(*b->b); return 0; PS_CATCH_INT }
int ps_batch_select_hard(ps_batch *b, uint64_t before, uint64_t *after) { PS_TRY batch_select_hard(*b->b, before, after); return 0; PS_CATCH_INT }
int ps_batch_select_easy(ps_batch *b, int threads) { PS_TRY b->b->ctx->host_threads = threads > 0 ? threads : 1; batch_select_easy(*b->b, threads); return 0; PS_CATCH_INT }
int ps_batch_locate(ps_batch *b) { PS_TRY batch_locate(*b->b); return 0; PS_CATCH_INT }
int ps_batch_run(ps_batch *b, int threads)
{
    PS_TRY
        b->b->ctx->host_threads = threads > 0 ? threads : 1;
        batch_search(*b->b);
        batch_select_hard(*b->b, 0, nullptr);
        batch_select_easy(*b->b, threads);
        batch_locate(*b->b);
        return 0;
    PS_CATCH_INT
}
int ps_batch_write_sam(ps_batch *b, const char *path, int with_header, int threads)
{
    PS_TRY
        batch_write_sam(*b->b, path, with_header != 0, "@PG\tID:parasuite-hip\tPN:parasuite-hip\tVN:0.1", threads); return 0;
    PS_CATCH_INT
}
int ps_batch_n_aln(ps_batch *b, int32_t *out, int64_t cap)
{
    PS_TRY
        Batch &B = *b->b;
        for (int64_t g = 0; g < B.rs.n && g < cap; ++g) { int n; B.alns_of(g, n); out[g] = n; }
        return 0;
    PS_CATCH_INT
}
int64_t ps_batch_alns(ps_batch *b, int64_t read, ps_aln *out, int64_t cap)
{
    try {
        int n; const AlnRec *a = b->b->alns_of(read, n);
        for (int j = 0; j < n && j < cap; ++j) std::memcpy(&out[j], &a[j], sizeof(ps_aln));
        return n;
    } catch (const std::exception &e) { fail(e.what()); return -1; }
}
int ps_batch_hits(ps_batch *b, ps_hit *out, int64_t cap)
{
    PS_TRY
        Batch &B = *b->b;
        for (int64_t g = 0; g < B.rs.n && g < cap; ++g) {
            Hit h; B.hit_of(g, h); ps_hit &o = out[g];
            o.pos = h.type ? h.pos : -1; o.sa = h.sa; o.type = h.type; o.strand = h.strand; o.mapq = h.mapq; o.n_mm = h.n_mm; o.n_gapo = h.n_gapo;
            o.n_gape = h.n_gape; o.ref_shift = h.ref_shift; o.score = h.score; o.c1 = h.c1; o.c2 = h.c2; o.n_cigar = h.n_cigar; o.n_multi = h.n_multi;
            std::memset(o.cigar, 0, sizeof o.cigar); std::memcpy(o.cigar, h.cigar, sizeof h.cigar);
        }
        return 0;
    PS_CATCH_INT
}
int64_t ps_ctx_read_iters(ps_ctx *x, uint32_t *out, int64_t cap)
{
    const int64_t n = (int64_t)x->c.read_iters.size();
    for (int64_t i = 0; i < n && i < cap; ++i) out[i] = x->c.read_iters[i];
    return n;
}
int ps_batch_timing(ps_batch *b, ps_timing *o)
{
    PS_TRY
        const Timing &t = b->b->tm;
        o->ms_width = t.ms_width; o->ms_backtrack = t.ms_backtrack; o->ms_compact = t.ms_compact; o->ms_select = t.ms_select;
        o->ms_sa2pos = t.ms_sa2pos; o->ms_refine = t.ms_refine; o->ms_host_post = t.ms_host_post; o->ms_total = t.ms_total;
        o->ms_classify = t.ms_classify; o->ms_rows = t.ms_rows; o->ms_sel_hard = t.ms_sel_hard; o->ms_sel_easy = t.ms_sel_easy;
        o->n_width_launches = t.n_width_launches; o->n_backtrack_launches = t.n_backtrack_launches;
        o->n_overflow_tier1 = b->b->n_overflow[1]; o->n_overflow_tier2 = b->b->n_overflow[2];
        o->bt_begin_ms = t.bt_begin_ms; o->bt_end_ms = t.bt_end_ms;
        return 0;
    PS_CATCH_INT
}
int ps_batch_kstats(ps_batch *b, int which, ps_kstats *o)
{
    PS_TRY
        const KStats &k = which == 0 ? b->b->st_width : (which == 1 ? b->b->st_backtrack : b->b->st_sa2pos);
        o->occ_pairs = k.occ_pairs; o->occ_same_blk = k.occ_same_blk; o->nodes = k.nodes; o->pushes = k.pushes; o->pops = k.pops;
        o->lf_steps = k.lf_steps; o->iters = k.iters; o->exact_steps = k.exact_steps; return 0;
    PS_CATCH_INT
}

// The whole `map` step behind one call.  Stages run side by side on pieces of the input (whole records; the file is
// streamed, a window at a time): a parser thread (parse, bin, 2-bit pack), GPU workers (upload, search, samse stage) and a
// writer thread formatting SAM in input order.
// Devices: the first PARASUITE_GPUS devices (default 1), or the list in PARASUITE_GPU_IDS.  Every device holds ONE copy of
// the index: the first loads the files, the others receive the three blobs from it over xGMI (hipMemcpyPeerAsync).  Every
// device has PS_WORKERS_PER_GPU workers (default 1; 2 is allowed, and a device named twice in PARASUITE_GPU_IDS gets two), each
// with its own stream and workspace.  Two workers on one device were measured SLOWER end to end (4.6 s against 4.2 s for
// 10 M reads): two persistent search kernels share the CUs evenly instead of one refilling the other's tail, the later
// stages of one piece starve under the other's kernel, and the second 69 GB workspace costs its allocation.  The piece size
// follows from the input: at least two pieces per worker, at most 400 MB of text each; PS_CHUNK_MB states a fixed size.
// Pieces go to whichever worker is free; the one sequential thing, the tie-break stream, is handed from piece to piece in
// input order (only the reads whose draw count is data dependent sit on that chain), so the SAM does not depend on the cut,
// on the number of workers or on the number of devices.  A finished piece gives its device memory back at once and at most
// a few finished pieces wait for the writer: memory does not grow with the input.
static const char *const PS_PG_LINE = "@PG\tID:parasuite-hip\tPN:parasuite-hip\tVN:0.1";
namespace {
struct ProfileSink { int min_mapq = 0, max_len = 0; std::string prefix; };    // ps_map_profiled: the first pass also counts its error profile
struct BamOut { int min_mapq = 0; bool sort = false, index = false; int level = 1; BamStats *stats = nullptr; };   // ps_map_to_bam: out_sam names a BAM file
}
static int map_core(int threads, const char *mm, const char *error_profile, const char *indel_profile,
                    const char *ref_fa, const char *fastq, const char *out_sam, const ProfileSink *sink, const BamOut *bam = nullptr)
{
    PS_TRY
        const bool verbose = std::getenv("PS_VERBOSE") != nullptr;
        const auto t_begin = std::chrono::steady_clock::now();
        auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
        const int nthr = threads > 0 ? threads : 1;
        // ---- devices and workers
        std::vector<int> devs, dev_workers;              // distinct devices in the order named; workers on each
        int per_dev = 1;
        if (const char *e = std::getenv("PS_WORKERS_PER_GPU")) per_dev = std::max(1, std::min(std::atoi(e), (int)Ctx::N_WORK));
        {
            std::vector<int> named;
            if (const char *e = std::getenv("PARASUITE_GPU_IDS")) { for (const char *p = e; *p;) { named.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; } }
            else {
                int want = 1, have = 1;
                if (const char *e = std::getenv("PARASUITE_GPUS")) want = std::max(1, std::atoi(e));
                if (want > 1 && (hipGetDeviceCount(&have) != hipSuccess || have < 1)) return fail("no HIP device available");   // one device: found out (loudly) when the worker attaches it
                for (int g = 0; g < std::min(want, have); ++g) named.push_back(g);
            }
            if (named.empty()) named.push_back(0);
            for (int d : named) {
                size_t k = 0;
                while (k < devs.size() && devs[k] != d) ++k;
                if (k == devs.size()) { devs.push_back(d); dev_workers.push_back(0); }
                ++dev_workers[k];
            }
            for (int &w : dev_workers) w = std::min((int)Ctx::N_WORK, std::max(w, per_dev));
        }
        const int G = (int)devs.size();
        int n_workers = 0; for (int w : dev_workers) n_workers += w;
        // ---- piece size from the input
        // Few, large pieces: every search launch ends with its longest read (~0.25 s of a launch are that, whatever its size: a 1.25 M-read
        // launch takes 0.37 s, 10 M reads in one 1.1 s).  The parser hands over what it has when a worker WAITS for work (the first piece
        // as soon as the index is resident) but not less than 20 % of the input, and otherwise lets a piece grow to 1 GB; with several
        // workers a piece is at most 1/(2 x workers) of the input, so that all of them get some.
        size_t chunk_bytes = (size_t)1 << 30, first_bytes = 0, hungry_min = (size_t)128 << 20;
        if (const char *e = std::getenv("PS_CHUNK_MB")) { chunk_bytes = (size_t)std::max(1, std::atoi(e)) << 20; hungry_min = chunk_bytes; }   // stated: taken as it is
        else {
            FILE *f = std::fopen(fastq, "rb");
            if (f) {
                if (fseeko(f, 0, SEEK_END) == 0) {
                    const off_t sz = ftello(f);
                    if (sz > 0) {
                        if (n_workers > 1) chunk_bytes = std::min(chunk_bytes, std::max<size_t>((size_t)16 << 20, ((size_t)sz + 2 * (size_t)n_workers - 1) / (2 * (size_t)n_workers) + ((size_t)64 << 10)));   // + slack: cuts fall behind whole records, the last piece must not be a few reads
                        hungry_min = std::min(chunk_bytes, std::max(hungry_min, (size_t)sz / 5));
                        // BAM out: compressing the records (2.3 s per 10 M reads at zlib level 1, 16 threads) is the slowest stage and can only
                        // start on a piece the GPU has finished -- four pieces, so that it starts early (3.5 -> 3.0 s per 10 M reads)
                        if (bam) { chunk_bytes = std::min(chunk_bytes, std::max<size_t>((size_t)64 << 20, (size_t)sz / 4 + ((size_t)64 << 10))); hungry_min = std::min(hungry_min, chunk_bytes / 2); }
                    }
                }
                std::fclose(f);
            }
        }
        if (const char *e = std::getenv("PS_HUNGRY_MIN_MB")) hungry_min = (size_t)std::max(1, std::atoi(e)) << 20;
        if (const char *e = std::getenv("PS_FIRST_MB")) first_bytes = (size_t)std::max(1, std::atoi(e)) << 20;      // first piece (the following ones double up to the piece size)
        struct Piece { int64_t seq = 0; std::unique_ptr<Batch> b; };
        Chan<Piece> parsed;
        parsed.cap = (size_t)std::max(2, n_workers);
        // the contexts (stream, options) exist before any index is loaded: the parser stage needs the cost model to bin
        // and pack the reads, not the index
        std::vector<ps_ctx *> xs((size_t)G, nullptr);
        auto close_all = [&]() { for (ps_ctx *c : xs) if (c) ps_ctx_close(c); };
        for (int g = 0; g < G; ++g) {
            xs[g] = new_ctx(devs[g], false);               // the device side is attached by the device's first worker, beside the parser
            if (error_profile && error_profile[0] ? ps_ctx_set_profile(xs[g], error_profile, indel_profile, mm)
                                                   : ps_ctx_set_stock(xs[g], mm && mm[0] ? mm : "0.04")) { const std::string m = g_err; close_all(); return fail(m); }
            xs[g]->c.host_threads = nthr;
            xs[g]->c.n_work = dev_workers[g];
        }
        std::mutex mu; std::condition_variable cv;       // guards: failure, the tie-break chain, the finished pieces, index hand-out
        bool failed = false; std::string msg;
        int64_t next_select = 0, write_next = 0; uint64_t draws = 0;
        std::map<int64_t, std::unique_ptr<Batch>> done; int workers_left = n_workers;
        const size_t done_cap = (size_t)n_workers + 2;   // finished pieces that may wait for the writer
        std::vector<int> index_state((size_t)G, 0);      // 0 not there, 1 resident
        std::vector<int> attached((size_t)G, 0);         // the device side of the context exists (made by the device's first worker)
        auto fail_all = [&](const std::string &m) { { std::lock_guard<std::mutex> l(mu); if (!failed) { failed = true; msg = m; } } cv.notify_all(); parsed.abort(); };
        double t_parse = 0, t_write = 0, t_release = 0, t_index = 0, t_index_all = 0, t_profile = 0; std::vector<double> t_gpu((size_t)n_workers, 0.0);
        int64_t n_reads = 0, n_pieces = 0;
        Trash::trash().collect();                        // what an earlier call left to be freed
        // ---- parser (starts at once)
        std::thread parser([&]() {
            try {
                int64_t seq = 0;
                int pthr = nthr;                                       // all of them: the GPU waits for the first piece, and sharing the cores with the writer later cost nothing measurable (2.78-2.90 -> 2.67-2.80 s per 10 M reads against half of them)
                if (const char *e = std::getenv("PS_PARSE_THREADS")) pthr = std::max(1, std::atoi(e));
                const std::function<bool()> hungry = [&]() { return parsed.hungry(); };
                load_reads_chunked(fastq, pthr, chunk_bytes, [&](ReadSet &&rs) {
                    Piece p; p.seq = seq++; p.b = batch_prepare(&xs[0]->c, std::move(rs), pthr);     // host only
                    parsed.push(std::move(p));
                }, first_bytes, &hungry, hungry_min);
                t_parse = since();
            } catch (const std::exception &e) { fail_all(e.what()); }
            parsed.close();
        });
        // ---- writer: pieces in input order
        std::thread writer([&]() {
            try {
                bool first = true;
                SamScratch scratch;
                if (!bam) { const int fd = ::open(out_sam, O_WRONLY | O_CREAT | O_TRUNC, 0644); if (fd >= 0) ::close(fd); }    // an output file that exists is emptied now, while this thread has nothing to do: giving back 2 GB of cached pages takes 0.3 s
                std::unique_ptr<BamSink> bsink;                        // ps_map_to_bam: records go out as BAM, no SAM text at all
                std::unique_ptr<ProfileAccum> accum;                   // on the first device, whose index is resident before any piece is finished
                for (;;) {
                    std::unique_ptr<Batch> b;
                    {
                        std::unique_lock<std::mutex> l(mu);
                        cv.wait(l, [&] { return failed || done.count(write_next) || (workers_left == 0 && done.empty()); });
                        if (failed) return;
                        auto it = done.find(write_next);
                        if (it == done.end()) break;                       // all workers finished and nothing is left
                        b = std::move(it->second); done.erase(it);
                    }
                    const auto t0 = std::chrono::steady_clock::now();
                    if (bam) {
                        if (!bsink) {
                            std::vector<std::pair<std::string, uint32_t>> refs;
                            for (const Contig &ct : b->ctx->ix.ref.contigs) refs.emplace_back(ct.name, (uint32_t)ct.len);
                            bsink.reset(new BamSink(sam_header(b->ctx->ix.ref, PS_PG_LINE), refs, out_sam, bam->sort, bam->index, nthr, bam->level));
                        }
                        std::vector<std::string> enc; std::vector<std::vector<BamRec>> recs;
                        batch_bam_records(*b, bam->min_mapq, nthr, enc, recs);
                        bsink->add(enc, recs, (uint64_t)b->rs.n);
                    } else batch_write_sam(*b, out_sam, first, PS_PG_LINE, nthr, !first, &scratch);
                    t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (sink) {                                        // the same records, straight from memory, into the profile histograms
                        const auto tp = std::chrono::steady_clock::now();
                        if (!accum) accum.reset(new ProfileAccum(xs[0]->c.device, xs[0]->c.ix, sink->max_len));
                        ProfRecords pr;
                        batch_profile_records(*b, sink->min_mapq, nthr, pr);
                        accum->add(pr);
                        t_profile += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count();
                    }
                    first = false;
                    { std::lock_guard<std::mutex> l(mu); ++write_next; }
                    cv.notify_all();
                    const auto t1 = std::chrono::steady_clock::now();
                    { Batch *q = b.release(); Trash::trash().add(std::thread([q]() { delete q; })); }   // pinned record buffers, the reads (~40 ms per piece): released on a thread of its own, neither on the GPU worker's time nor on the writer's
                    t_release += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
                }
                if (first) {                      // no reads at all: the header alone, as upstream's samse prints it before its read loop
                    { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return failed || index_state[0] == 1; }); if (failed) return; }
                    const std::string h = sam_header(xs[0]->c.ix.ref, PS_PG_LINE);
                    if (bam) {
                        std::vector<std::pair<std::string, uint32_t>> refs;
                        for (const Contig &ct : xs[0]->c.ix.ref.contigs) refs.emplace_back(ct.name, (uint32_t)ct.len);
                        bsink.reset(new BamSink(h, refs, out_sam, bam->sort, bam->index, nthr, bam->level));
                    }
                    FILE *f = bam ? nullptr : std::fopen(out_sam, "wb");
                    if (bam) { /* the header-only BAM is written by finish() below */ } else {
                    if (!f) throw Error(std::string("cannot write ") + out_sam);
                    const bool ok = std::fwrite(h.data(), 1, h.size(), f) == h.size();
                    if (std::fclose(f) != 0 || !ok) throw Error(std::string("short write on ") + out_sam);
                    }
                }
                if (bsink) { const auto t0 = std::chrono::steady_clock::now(); bsink->finish(bam->stats); bsink.reset(); t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
                if (sink) {
                    ProfileCounts pc;
                    if (accum) accum->finish(pc);
                    else { pc.max_len = sink->max_len; pc.conv.assign((size_t)sink->max_len * 16, 0); pc.ins.assign((size_t)sink->max_len, 0); pc.del.assign((size_t)sink->max_len, 0); }
                    error_profile_write(pc, sink->prefix);
                    accum.reset();                                      // before the contexts (and their devices' memory) go
                }
            } catch (const std::exception &e) { fail_all(e.what()); }
        });
        // ---- the workers
        auto worker = [&](int g, int j, int slot) {
            try {
                Ctx &c = xs[g]->c;
                if (j == 0) { ctx_attach_device(xs[g]); { std::lock_guard<std::mutex> l(mu); attached[g] = 1; } cv.notify_all(); }
                else { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return failed || attached[g] == 1; }); if (failed) return; }
                require_device(c.device);
                // The worker's big allocations (69 GB of stack slices + the large slots) are made NOW, on a thread of their own, while the
                // index loads and the parser works on the first piece: a hipMalloc that is handed memory another call or process has
                // just freed waits for the driver to clear it (seconds for this size, tools/microbench_malloc) -- behind the index load
                // that wait is hidden, in front of the first search launch (where the first ws_get used to make it) it is not.  With
                // several workers on one device it also keeps a worker's allocation from waiting for another worker's running kernel.
                std::thread reserve([&c, j]() { try { reserve_search_workspace(&c, j); } catch (...) {} });
                struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } reserve_joiner{reserve};
                if (j == 0) {                                          // this device's index: from the files, or from the first device
                    if (g == 0) { index_load(ref_fa, c.ix, c.stream); t_index = since(); }
                    else {
                        { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return failed || index_state[0] == 1; }); if (failed) return; }
                        index_clone(xs[0]->c.ix, xs[0]->c.device, c.ix, c.device, c.stream);
                    }
                    { std::lock_guard<std::mutex> l(mu); index_state[g] = 1; t_index_all = since(); }
                    cv.notify_all();
                } else { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return failed || index_state[g] == 1; }); if (failed) return; }
                if (reserve.joinable()) reserve.join();
                Piece p;
                while (parsed.pop(p)) {
                    { std::lock_guard<std::mutex> l(mu); if (failed) return; n_reads += p.b->rs.n; ++n_pieces; }
                    const auto t0 = std::chrono::steady_clock::now();
                    Batch &b = *p.b;
                    b.ctx = &c;                                        // the piece was packed with the (identical) options of context 0
                    b.work_index = j;
                    batch_upload(b);
                    const double w_up = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    batch_search(b);
                    const double w_search = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - w_up;
                    {                                                   // the tie-break stream: pieces take their turn in input order
                        std::unique_lock<std::mutex> l(mu);
                        cv.wait(l, [&] { return failed || next_select == p.seq; });
                        if (failed) return;
                        uint64_t after = 0;
                        batch_select_hard(b, draws, &after);
                        draws = after; ++next_select;
                    }
                    cv.notify_all();
                    batch_select_easy(b, nthr);
                    batch_locate(b);
                    b.release_device();                                // what is left to do (SAM text) reads host memory only
                    if (verbose) {
                        const Timing &t = b.tm;
                        std::fprintf(stderr, "[parasuite-hip]   piece %lld on device %d worker %d: %lld reads; upload %.0f ms, search stage %.0f ms wall (width %.0f backtrack %.0f classify %.0f), select %.0f+%.0f sa2pos %.0f refine %.0f host_post %.0f ms\n",
                                     (long long)p.seq + 1, c.device, j, (long long)b.rs.n, 1e3 * w_up, 1e3 * w_search, t.ms_width, t.ms_backtrack, t.ms_classify, t.ms_sel_hard, t.ms_sel_easy, t.ms_sa2pos, t.ms_refine, t.ms_host_post);
                    }
                    t_gpu[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    {
                        // at most done_cap finished pieces wait for the writer -- but the piece the writer wants next always gets in
                        std::unique_lock<std::mutex> l(mu);
                        cv.wait(l, [&] { return failed || done.size() < done_cap || p.seq == write_next; });
                        if (failed) return;
                        done[p.seq] = std::move(p.b);
                    }
                    cv.notify_all();
                }
            } catch (const std::exception &e) { fail_all(e.what()); }
        };
        auto worker_exit = [&](int g, int j, int slot) { worker(g, j, slot); { std::lock_guard<std::mutex> l(mu); --workers_left; } cv.notify_all(); };
        bool have_index = true;
        try {
            if (!index_files_exist(ref_fa)) {         // the Java probes <ref>.bwt and indexes first; be lenient if it did not
                ctx_attach_device(xs[0]);
                Index tmp; index_build(ref_fa, tmp, xs[0]->c.stream); index_save(tmp, ref_fa);
            }
        } catch (const std::exception &e) { have_index = false; fail_all(e.what()); }
        std::vector<std::thread> workers;
        if (have_index) { int slot = 0; for (int g = 0; g < G; ++g) for (int j = 0; j < dev_workers[g]; ++j) workers.emplace_back(worker_exit, g, j, slot++); }
        else { std::lock_guard<std::mutex> l(mu); workers_left = 0; }
        for (auto &t : workers) t.join();
        const double t_workers = since();
        cv.notify_all();
        parsed.abort();                           // a parser still waiting to hand over a piece must not wait forever
        // The devices' memory (index, 69-GB workspaces: ~0.1 s of hipFree) goes back while the writer formats the last piece, which
        // reads host memory only; the error-profile pass keeps its device (ProfileAccum counts on it until the writer is done).
        std::thread early_release;
        double t_release_dev = 0;
        if (!sink) early_release = std::thread([&]() { for (ps_ctx *c : xs) if (c) { try { ctx_release_device(c->c); } catch (...) {} } t_release_dev = since(); });
        parser.join(); writer.join();
        const double t_written = since();
        if (early_release.joinable()) early_release.join();
        done.clear();
        // what is left of the contexts (streams, events, the mapped packed text: 0.05 s) goes the way of the written pieces when the
        // device memory has been given back already; a failed call and the error-profile pass close them here
        if (!failed && !sink) { std::vector<ps_ctx *> gone(xs); for (auto &c : xs) c = nullptr; Trash::trash().add(std::thread([gone]() { for (ps_ctx *c : gone) if (c) ps_ctx_close(c); })); }
        else close_all();
        const double t_closed = since();
        if (failed) return fail(msg);
        if (verbose) {
            double busy = 0; for (double v : t_gpu) busy += v;
            std::fprintf(stderr, "[parasuite-hip] ps_map: %lld reads in %lld piece(s) of <= %.0f MB, %d device(s) x %d worker(s), %.3f s; index resident after %.3f s (all devices %.3f s), "
                                 "parser done after %.3f s, GPU stages busy %.3f s (summed over workers) and done after %.3f s, SAM writer busy %.3f s (+ %.3f s handing pieces back, %.3f s error profile) and done after %.3f s, device memory released after %.3f s, contexts closed after %.3f s\n", (long long)n_reads, (long long)n_pieces, chunk_bytes / 1048576.0,
                                 G, dev_workers[0], since(), t_index, t_index_all, t_parse, busy, t_workers, t_write, t_release, t_profile, t_written, t_release_dev, t_closed);
        }
        return 0;
    PS_CATCH_INT
}

int ps_map(int threads, const char *mm, const char *error_profile, const char *indel_profile,
           const char *ref_fa, const char *fastq, const char *out_sam)
{
    return map_core(threads, mm, error_profile, indel_profile, ref_fa, fastq, out_sam, nullptr);
}
// ps_map + the error profile of its own alignments (those with MAPQ >= min_mapq: what the filtered BAM of the pass would
// hold), counted from the records in memory while the SAM is being written: <profile_prefix>.errorprofile / .indelprofile
int ps_map_profiled(int threads, const char *mm, const char *error_profile, const char *indel_profile,
                    const char *ref_fa, const char *fastq, const char *out_sam, int min_mapq, int max_read_len, const char *profile_prefix)
{
    if (!profile_prefix || !profile_prefix[0]) return fail("ps_map_profiled: no output prefix for the profile files");
    if (max_read_len < 1 || max_read_len > 4096) return fail("error profile: maximum read length out of range");
    ProfileSink sink; sink.min_mapq = min_mapq; sink.max_len = max_read_len; sink.prefix = profile_prefix;
    return map_core(threads, mm, error_profile, indel_profile, ref_fa, fastq, out_sam, &sink);
}

// ps_map with the records going straight into a BAM file: what PARAsuiteMapping.java:102-152 makes of <prefix>.sam with three
// samtools calls (view -bS, view -q, and -- Mapping.java:85-108 -- sort + index), without the 2 GB of SAM text in between.  Records
// with MAPQ < min_mapq are left out; sort_by_coordinate / write_index as in ps_sam_to_bam.  Unsorted output is compressed and written
// piece by piece while later pieces are searched.  zlib level 1 by default (the BAM is 10 % larger than at samtools' level 6 and the call
// 0.7 s shorter per 10 M reads: compression, not mapping, is what the host spends its time on); PS_BAM_LEVEL=6 for samtools' own.
int ps_map_to_bam(int threads, const char *mm, const char *error_profile, const char *indel_profile,
                  const char *ref_fa, const char *fastq, const char *out_bam, int min_mapq, int sort_by_coordinate, int write_index, ps_bam_stats *st)
{
    if (write_index && !sort_by_coordinate) return fail("a .bai index needs coordinate-sorted output");
    BamStats s;
    BamOut bo; bo.min_mapq = min_mapq; bo.sort = sort_by_coordinate != 0; bo.index = write_index != 0; bo.stats = &s;
    if (const char *e = std::getenv("PS_BAM_LEVEL")) bo.level = std::atoi(e);
    const int rc = map_core(threads, mm, error_profile, indel_profile, ref_fa, fastq, out_bam, nullptr, &bo);
    if (rc == 0 && st) { st->n_in = s.n_in; st->n_out = s.n_out; st->bam_bytes = s.bam_bytes; }
    return rc;
}

// page-locked host buffers the library keeps between calls (ps_pipeline.h, PinBuf): given back to the system
void ps_release_host_cache(void) { try { Trash::trash().collect(); pin_cache_release(); } catch (...) {} }

// host-only: parse reads the way ps_map does (whole file on `threads` threads, or streamed in windows of chunk_bytes) and
// summarise what came out -- {reads, bases, order-sensitive hash of names/sequences/qualities, pieces}
int ps_parse_check(const char *reads_path, int threads, uint64_t chunk_bytes, uint64_t out[4])
{
    PS_TRY
        uint64_t n = 0, bases = 0, h = 1469598103934665603ull, pieces = 0;
        auto mix = [&](const void *p, size_t len) { const unsigned char *c = (const unsigned char *)p; for (size_t i = 0; i < len; ++i) { h ^= c[i]; h *= 1099511628211ull; } h ^= 0xff; h *= 1099511628211ull; };
        auto eat = [&](const ReadSet &rs) {
            ++pieces;
            for (int64_t i = 0; i < rs.n; ++i) {
                size_t nl; const char *nm = rs.name(i, nl);
                mix(nm, nl); mix(rs.seq.data() + rs.off[i], (size_t)rs.len[i]); mix(rs.qual.data() + rs.off[i], (size_t)rs.len[i]);
                bases += (uint64_t)rs.len[i];
            }
            n += (uint64_t)rs.n;
        };
        if (chunk_bytes == 0) { ReadSet rs; load_reads(reads_path, rs, threads); eat(rs); }
        else if (const char *e = std::getenv("PS_PARSE_CHECK_HUNGRY")) {       // tests: a consumer that always waits, as ps_map's GPU worker does at the start: pieces go out at `e` bytes
            const std::function<bool()> hungry = []() { return true; };
            load_reads_chunked(reads_path, threads, (size_t)chunk_bytes, [&](ReadSet &&rs) { eat(rs); }, 0, &hungry, (size_t)std::max(1, std::atoi(e)));
        }
        else load_reads_chunked(reads_path, threads, (size_t)chunk_bytes, [&](ReadSet &&rs) { eat(rs); });
        out[0] = n; out[1] = bases; out[2] = h; out[3] = pieces;
        return 0;
    PS_CATCH_INT
}

// Error-profile estimation from a mapping (the stage between the two passes of a --refine run, Main.java:320-340): what
// `new ErrorProfiling(mapping, reference, maxReadLength).inferErrorProfile(false, false)` writes for the mapper.
int ps_error_profile(const char *mapping_sam_or_bam, const char *ref_fa, int max_read_len, const char *out_prefix)
{
    PS_TRY
        ProfileCounts c;
        int dev = 0;
        if (const char *e = std::getenv("PARASUITE_GPU_IDS")) dev = std::atoi(e);
        error_profile_count(mapping_sam_or_bam, ref_fa, max_read_len, dev, 8, c);
        error_profile_write(c, out_prefix && out_prefix[0] ? out_prefix : mapping_sam_or_bam);
        if (std::getenv("PS_VERBOSE"))
            std::fprintf(stderr, "[parasuite-hip] ps_error_profile: %llu records, %llu counted (%llu unmapped, %llu duplicate, %llu without position, %llu with indels, %llu skipped)\n",
                         c.n_records, c.n_processed, c.n_unmapped, c.n_duplicate, c.n_start_zero, c.n_indel_reads, c.n_skipped);
        return 0;
    PS_CATCH_INT
}

int ps_sam_to_bam(const char *sam, const char *bam, int min_mapq, int sort_by_coordinate, int write_index, int threads, ps_bam_stats *st)
{
    PS_TRY
        BamStats s;
        sam_to_bam(sam, bam, min_mapq, sort_by_coordinate != 0, write_index != 0, threads, &s);
        if (st) { st->n_in = s.n_in; st->n_out = s.n_out; st->bam_bytes = s.bam_bytes; }
        return 0;
    PS_CATCH_INT
}

int ps_bam_view(const char *in_bam, const char *out_bam, int min_mapq, int threads, ps_bam_stats *st)
{
    PS_TRY
        BamStats s; bam_view(in_bam, out_bam, min_mapq, threads, &s);
        if (st) { st->n_in = s.n_in; st->n_out = s.n_out; st->bam_bytes = s.bam_bytes; }
        return 0;
    PS_CATCH_INT
}
int ps_bam_sort(const char *in_bam, const char *out_bam, int by_name, int threads, ps_bam_stats *st)
{
    PS_TRY
        BamStats s; bam_sort(in_bam, out_bam, by_name != 0, threads, &s);
        if (st) { st->n_in = s.n_in; st->n_out = s.n_out; st->bam_bytes = s.bam_bytes; }
        return 0;
    PS_CATCH_INT
}
int ps_bam_index(const char *bam, int threads)
{
    PS_TRY
        bam_index(bam, threads); return 0;
    PS_CATCH_INT
}

}  // extern "C"
