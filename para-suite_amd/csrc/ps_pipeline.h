// ps_pipeline.h -- batch pipeline behind the C ABI (host orchestration of the gfx950 kernels).
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include "ps_host.h"
#include "ps_bam.h"

namespace ps {

// std::vector without the zero fill of resize(): the parser's arrays (hundreds of MB per piece) are written once, by many threads
template <class T> struct DefaultInit : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInit<U>; };
    DefaultInit() noexcept {}
    template <class U> DefaultInit(const DefaultInit<U> &) noexcept {}
    template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using RawVec = std::vector<T, DefaultInit<T>>;

struct ReadSet {
    int64_t n = 0;
    RawVec<int32_t> len;
    RawVec<int64_t> off;             // n+1 offsets into seq / qual
    RawVec<uint8_t> seq;             // codes 0..3, 4 = N, read orientation
    RawVec<char> qual; bool has_qual = false;
    RawVec<char> names; RawVec<int64_t> name_off;            // n+1
    const char *name(int64_t i, size_t &l) const { l = (size_t)(name_off[i + 1] - name_off[i]); return names.data() + name_off[i]; }
};
void load_reads(const char *path, ReadSet &rs, int threads = 1); // FASTQ or FASTA
// The input in pieces of whole records, in order; sink(piece) may block.  The file is STREAMED in windows of <= 64 MB; a piece is the
// windows parsed so far and goes out when another window would take it over chunk_bytes (first_bytes for the first piece, doubling from
// there) or -- `hungry` given -- as soon as it holds hungry_min_bytes and hungry() says that the stage behind is waiting for work.
void load_reads_chunked(const char *path, int threads, size_t chunk_bytes, const std::function<void(ReadSet &&)> &sink, size_t first_bytes = 0,
                        const std::function<bool()> *hungry = nullptr, size_t hungry_min_bytes = 0);
void reads_from_codes(int64_t n, int len, const uint8_t *codes, ReadSet &rs);

static const int PS_HIT_CIGAR = 8;
struct Multi { int64_t pos; bwtint row; int32_t gap, mm, ref_shift, strand, n_cigar; uint32_t cigar[PS_MAX_CIGAR]; };
struct Hit {
    int64_t pos; bwtint sa;
    int32_t type, strand, mapq, n_mm, n_gapo, n_gape, ref_shift, score, c1, c2, n_cigar;
    int32_t multi_begin, n_multi;
    uint32_t cigar[PS_HIT_CIGAR];    // gapped main hits with more operations are rejected (needs max_gapo > 2)
};

// per-read records of the samse stage for reads finished on the device
struct SelRec { bwtint sa; int32_t c1, c2; uint8_t type, n_mm, n_gapo, n_gape; int8_t ref_shift; uint8_t score, pad[2]; };   // 24 B
struct FinRec { int64_t pos; uint8_t strand, mapq, type, pad[5]; };                                                        // 16 B
struct DevCigar { int64_t g; int32_t n; uint32_t c[PS_HIT_CIGAR]; };
// a read finished on the host: several best-score intervals (sequential tie-break), alternative hits, or a larger tier
struct SubRead { int64_t g = 0, easy_before = 0, hard_before = 0; uint8_t cls = 0; const AlnRec *alns = nullptr; int32_t n_alns = 0; Hit hit; };   // alns: into Batch::sub_alns or a Bin's overflow list

struct Timing {
    double ms_width = 0, ms_backtrack = 0, ms_compact = 0, ms_select = 0, ms_sa2pos = 0, ms_refine = 0, ms_host_post = 0, ms_total = 0, ms_classify = 0, ms_rows = 0, ms_sel_hard = 0, ms_sel_easy = 0;
    int n_width_launches = 0, n_backtrack_launches = 0;
    double bt_begin_ms = 0, bt_end_ms = 0;   // first search launch's start / last one's end on the context's clock (overlapping batches: the union)
};

// Page-locked host memory is dear to get and to give back (hundreds of MB per piece of a ps_map call: ~50 ms to lock, as much to
// unlock): buffers that are let go are kept, up to a bound, and handed to the next taker (the pieces of one call, the next
// call of the process).  pin_cache_release() gives everything kept back to the system.
void *pin_cache_take(size_t need, size_t &got);
void pin_cache_give(void *p, size_t bytes);
void pin_cache_release();
struct PinBuf {                     // page-locked host staging (D2H at PCIe rate instead of pageable copies)
    void *p = nullptr; size_t bytes = 0;
    PinBuf() {}
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { if (p) pin_cache_give(p, bytes); }
    void *get(size_t need) { if (need > bytes) { if (p) pin_cache_give(p, bytes); p = nullptr; bytes = 0; p = pin_cache_take(need, bytes); } return p; }
};

// One lane of work on the device: a stream with its own grow-only device workspace and pinned staging (hipMalloc /
// hipFree of GBs per call is slow).  A context has two, so that two batches can be in flight -- one batch's kernel tail
// and host-side stages overlap with the other's search kernel.
struct Work {
    hipStream_t stream = nullptr;
    std::map<std::string, DevBuf<uint8_t>> ws; std::map<std::string, PinBuf> pin;
    template <class T> T *ws_get(const std::string &name, size_t count) {
        DevBuf<uint8_t> &b = ws[name];
        const size_t need = count * sizeof(T);
        if (b.n < need) b.alloc(need + need / 8);
        return reinterpret_cast<T *>(b.p);
    }
    template <class T> T *pin_get(const std::string &name, size_t count) { return reinterpret_cast<T *>(pin[name].get(count * sizeof(T) + 64)); }
    Work() {}
    Work(const Work &) = delete;
    Work &operator=(const Work &) = delete;
    ~Work() { if (stream) (void)hipStreamDestroy(stream); }
};

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;    // index build / load
    hipEvent_t ref_event = nullptr;  // the context's clock: recorded once at creation
    Index ix;
    Options opt;
    int bt_blocks = 0;             // grid of the backtracking kernel (0 = 4 blocks per CU)
    uint32_t pool_cap[3] = {16384, 65535, 2000064};   // stack entries per lane: 16-byte narrow tiers, then the wide tier
    int aln_cap[3] = {8, 256, 65536};
    int aln_cap_short = 32;        // first tier, bins whose longest read has at most 40 bases (adapter-trimmed PAR-CLIP reads, the transcript pass): short reads
                                   // hit many places by chance -- 1.7 % of 18-40 bp reads list more than 8 intervals, 0.3 % more than 32; 0 = as aln_cap[0]
    bool want_kstats = false;      // search kernel with counters (KStats of the backtracking stage): measurement runs only
    bool want_read_iters = false; std::vector<uint32_t> read_iters;   // profiling aid: last launch's per-read profile (ps_types.h: BtArgs::read_iters), PS_RI_WORDS words per read, in the order the launch held the reads (a bin's leading-base order)
    int n_big = 4096;              // 1 MB stack slots a launch may hand to reads that outgrow their private slice
    int fetch_min = 8, hit_min = 1;    // batching hits costs more in idle lanes than it saves (measured)
    int host_threads = 8;
    static const int N_WORK = 4;
    std::unique_ptr<Work> work[N_WORK];
    int n_work = 1, next_work = 0; // lanes in use (1: every batch shares one workspace and stream); batches take them in turn
    Work *take_work();             // creates the lane's stream on first use
    Work *work_at(int i);          // lane i (ps_map: one per worker thread of the device); thread safe
    std::mutex work_mu;
    ~Ctx();
};

// Reads of one *cost class* -- lengths that get the same difference budget, seed rule and local-memory layout --
// share a bin and a launch; len is the longest of them, lens the length of every read (adapter-trimmed input has
// dozens of lengths: one launch per length would leave the device mostly idle).
struct Bin {
    int len = 0; Model md;
    bool ragged = false;                      // more than one length present
    std::vector<int32_t> lens; DevBuf<int32_t> d_lens;     // bin-local; d_lens allocated only when ragged
    std::vector<int32_t> ids;                 // global read index of every local read
    DevBuf<uint32_t> bases, nmask, w; DevBuf<uint8_t> cwb, cswb, status; DevBuf<AlnRec> alns; DevBuf<int32_t> n_aln;
    RawVec<uint32_t> h_bases, h_nmask;        // host copy (tier re-runs gather from it); every word is written by the packing threads, none zero-filled first
    DevBuf<int32_t> d_ids;                     // ids on the device (bin-local -> input order)
    DevBuf<AlnRec> d_alns; DevBuf<int32_t> d_n_aln; DevBuf<uint8_t> d_status; int aln_cap = 0;   // first-tier hit lists stay in HBM
    bool host_alns_valid = false;             // compact host copy, downloaded on demand (tests, ps_batch_alns)
    std::vector<int32_t> h_n_aln; std::vector<uint32_t> h_off; std::vector<AlnRec> h_alns;
    std::map<int32_t, std::vector<AlnRec>> overflow;                                          // reads that needed a larger tier
    int n_bw = 0, n_mw = 0;
};

struct Batch {
    Ctx *ctx = nullptr;
    Work *wk = nullptr;            // the lane of work this batch runs on
    int work_index = -1;           // >= 0: the lane to use (ps_map's workers); -1: the context's next one
    ReadSet rs;
    std::vector<Bin> bins;
    std::vector<int32_t> read_bin, read_local;
    // samse stage: classes and device records in input order; the host-finished subset
    DevBuf<uint8_t> d_class; DevBuf<uint32_t> d_eb, d_hb; DevBuf<bwtint> d_rows, d_pos; DevBuf<SelRec> d_sel; DevBuf<FinRec> d_fin;
    PinBuf p_class, p_sel, p_fin;
    uint8_t *h_class = nullptr; SelRec *h_sel = nullptr; FinRec *h_fin = nullptr;
    std::vector<SubRead> sub; std::vector<Multi> multis; std::vector<DevCigar> dev_cigars;
    std::vector<std::unique_ptr<PinBuf>> sub_alns;   // per bin: hit lists of the host-finished reads (stride aln_cap)
    int64_t n_class1 = 0, n_hard = 0;
    std::vector<uint64_t> hard_draws_cum;  // draws consumed by the several-best reads up to and including each
    uint64_t draws_in = 0, draws_out = 0;
    DevBuf<KStats> d_stats; KStats st_width{}, st_backtrack{}, st_sa2pos{};
    Timing tm;
    int64_t n_overflow[3] = {0, 0, 0};
    bool searched = false, selected_hard = false, selected = false, located = false;
    const AlnRec *alns_of(int64_t g, int &n);      // downloads the hit lists on first use
    void ensure_host_alns();
    void hit_of(int64_t g, Hit &h) const;
    void release_device();         // after batch_locate: everything batch_write_sam / hit_of read is in host memory
};

std::unique_ptr<Batch> batch_create(Ctx *ctx, ReadSet &&rs);   // bins by cost class, packs 2-bit, uploads (= prepare + upload)
std::unique_ptr<Batch> batch_prepare(Ctx *ctx, ReadSet &&rs, int threads);   // host half: needs ctx->opt only, no device call
void batch_upload(Batch &b);                                                  // device half
void batch_search(Batch &b);                                    // width + backtracking kernels (+ larger tiers), hit lists to host
void batch_select_hard(Batch &b, uint64_t draws_before, uint64_t *draws_after);
void batch_select_easy(Batch &b, int threads);
void ctx_release_device(Ctx &c);                               // every device allocation of the context (index blobs, lanes of work) freed; host-side reference data stays
void reserve_search_workspace(Ctx *ctx, int work_index);        // the big device allocations of a lane of work, ahead of its first search
void batch_locate(Batch &b);                                    // SA walk kernel, strand / MAPQ, banded DP of gapped hits
void batch_bam_records(const Batch &b, int min_mapq, int threads, std::vector<std::string> &enc, std::vector<std::vector<BamRec>> &recs);   // the located batch as BAM records (MAPQ >= min_mapq): one buffer per thread, buffers in input order
std::string sam_header(const RefSeq &ref, const char *pg_line);      // @SQ lines in FASTA order + the @PG line
// text buffers of the SAM writer, kept from piece to piece by ps_map: fresh ones are first touched page by page, and a page fault waits
// while another thread gives a finished piece's gigabytes back to the system (0.2 s per piece, measured)
struct SamScratch { std::vector<std::string> bufs[2]; };
void batch_write_sam(Batch &b, const char *path, bool header, const char *pg_line, int threads, bool append = false, SamScratch *scratch = nullptr);
// the located hits with MAPQ >= min_mapq as records for the error-profile kernel (what the MAPQ-filtered BAM of the first pass holds:
// PARAsuiteMapping.java:124-133 -> ErrorProfiling.java:145-172), appended to `out`; host memory only
void batch_profile_records(const Batch &b, int min_mapq, int threads, ProfRecords &out);

}  // namespace ps
