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
#include <mutex>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <cerrno>
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

Ctx::~Ctx() { if (ref_event) (void)hipEventDestroy(ref_event); if (stream) (void)hipStreamDestroy(stream); }
void ctx_release_device(Ctx &c)
{
    require_device(c.device);
    {
        std::lock_guard<std::mutex> l(c.work_mu);
        for (auto &w : c.work) w.reset();
    }
    c.ix.blocks.release(); c.ix.sa.release(); c.ix.pac.release(); c.ix.jump.release();     // the view keeps its (now dangling) device pointers: nothing may search with this context again
}
Work *Ctx::work_at(int w)
{
    std::lock_guard<std::mutex> l(work_mu);
    if (w < 0 || w >= (int)N_WORK) throw Error("internal: work lane out of range");
    if (!work[w]) { work[w].reset(new Work()); PS_HIP(hipStreamCreateWithFlags(&work[w]->stream, hipStreamNonBlocking)); }
    return work[w].get();
}
Work *Ctx::take_work()
{
    int w;
    { std::lock_guard<std::mutex> l(work_mu); w = next_work; next_work = (next_work + 1) % std::max(1, std::min(n_work, (int)N_WORK)); }
    return work_at(w);
}
namespace {
struct PinCache {
    std::mutex mu; std::vector<std::pair<void *, size_t>> kept; size_t held = 0;
    static constexpr size_t CAP = (size_t)3 << 30;            // bytes kept at most
};
PinCache &pin_cache() { static PinCache *c = new PinCache(); return *c; }   // never destroyed: buffers may be given back during exit
}
void *pin_cache_take(size_t need, size_t &got)
{
    PinCache &c = pin_cache();
    {
        std::lock_guard<std::mutex> l(c.mu);
        int best = -1;
        for (size_t i = 0; i < c.kept.size(); ++i)            // the smallest kept buffer that fits (the pieces of ps_map differ in size from call to call: a larger buffer than needed beats locking a new one)
            if (c.kept[i].second >= need && (best < 0 || c.kept[i].second < c.kept[(size_t)best].second)) best = (int)i;
        if (best >= 0) {
            void *p = c.kept[(size_t)best].first; got = c.kept[(size_t)best].second;
            c.held -= got; c.kept.erase(c.kept.begin() + best);
            return p;
        }
    }
    const size_t want = need + need / 8 + 4096;               // a little room: the next piece is rarely exactly as large
    void *p = nullptr;
    PS_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    got = want;
    return p;
}
void pin_cache_give(void *p, size_t bytes)
{
    if (!p) return;
    PinCache &c = pin_cache();
    {
        std::lock_guard<std::mutex> l(c.mu);
        if (c.held + bytes <= PinCache::CAP) { c.kept.emplace_back(p, bytes); c.held += bytes; return; }
    }
    (void)hipHostFree(p);
}
void pin_cache_release()
{
    PinCache &c = pin_cache();
    std::vector<std::pair<void *, size_t>> all;
    { std::lock_guard<std::mutex> l(c.mu); all.swap(c.kept); c.held = 0; }
    for (auto &e : all) (void)hipHostFree(e.first);
}

void Batch::release_device()
{
    for (Bin &bin : bins) {
        bin.bases.release(); bin.nmask.release(); bin.w.release(); bin.cwb.release(); bin.cswb.release(); bin.status.release();
        bin.alns.release(); bin.n_aln.release(); bin.d_lens.release(); bin.d_ids.release();
        bin.d_alns.release(); bin.d_n_aln.release(); bin.d_status.release();
    }
    d_class.release(); d_eb.release(); d_hb.release(); d_rows.release(); d_pos.release(); d_sel.release(); d_fin.release(); d_stats.release();
    searched = false;              // the batch can be written, not searched again
}

// ------------------------------------------------------------- read input ----
static inline uint8_t code_of(int ch)
{
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                  case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

// FASTQ / FASTA; read name = header up to the first white space, a trailing /1 or /2 removed
// one parser pass over buf[i0, i1): appends to rs (offsets relative to rs's own arrays)
static void parse_reads_range(const char *buf, size_t i0, size_t i1, ReadSet &rs, bool &any_qual)
{
    struct Lut { uint8_t v[256]; Lut() { for (int c = 0; c < 256; ++c) v[c] = code_of(c); } };
    static const Lut lut_obj;                       // initialised once, safely, however many parser threads arrive
    const uint8_t *lut = lut_obj.v;
    size_t i = i0;
    const size_t n = i1;
    while (i < n) {
        while (i < n && buf[i] != '@' && buf[i] != '>') ++i;
        if (i >= n) break;
        const bool fq = buf[i] == '@';
        size_t s = ++i;
        while (i < n && !std::isspace((unsigned char)buf[i])) ++i;
        size_t nl = i - s;
        if (nl > 2 && buf[s + nl - 2] == '/' && (buf[s + nl - 1] == '1' || buf[s + nl - 1] == '2')) nl -= 2;
        rs.names.insert(rs.names.end(), buf + s, buf + s + nl);
        rs.name_off.push_back((int64_t)rs.names.size());
        const char *eol = (const char *)std::memchr(buf + i, '\n', n - i);
        i = eol ? (size_t)(eol - buf) + 1 : n;
        const size_t before = rs.seq.size();
        const char stop = fq ? '+' : '>';
        while (i < n && buf[i] != stop) {                         // sequence lines
            eol = (const char *)std::memchr(buf + i, '\n', n - i);
            size_t e = eol ? (size_t)(eol - buf) : n, e2 = e;
            while (e2 > i && !std::isgraph((unsigned char)buf[e2 - 1])) --e2;     // trailing CR / blanks
            const size_t at = rs.seq.size();
            rs.seq.resize(at + (e2 - i));
            for (size_t j = i; j < e2; ++j) rs.seq[at + (j - i)] = lut[(unsigned char)buf[j]];
            i = e < n ? e + 1 : n;
        }
        const int32_t len = (int32_t)(rs.seq.size() - before);
        rs.len.push_back(len);
        rs.off.push_back((int64_t)rs.seq.size());
        const size_t qbefore = rs.qual.size();
        if (fq && i < n) {
            eol = (const char *)std::memchr(buf + i, '\n', n - i);
            i = eol ? (size_t)(eol - buf) + 1 : n;
            while (i < n && (int32_t)(rs.qual.size() - qbefore) < len) {           // quality lines
                eol = (const char *)std::memchr(buf + i, '\n', n - i);
                size_t e = eol ? (size_t)(eol - buf) : n, e2 = e;
                while (e2 > i && !std::isgraph((unsigned char)buf[e2 - 1])) --e2;
                size_t take = e2 - i, room = (size_t)len - (rs.qual.size() - qbefore);
                if (take > room) take = room;
                rs.qual.insert(rs.qual.end(), buf + i, buf + i + take);
                i = e < n ? e + 1 : n;
            }
            any_qual = true;
        }
        rs.qual.resize(qbefore + (size_t)len, '!');
        ++rs.n;
    }
}

// ---- record boundaries ----------------------------------------------------------------------------------------
// A piece of the input must begin at a record.  A line that starts with '@' need not be a header (a quality string may
// start with '@'), so a candidate is VERIFIED by walking the record: header, sequence lines up to the '+' line, quality
// lines holding exactly as many characters as the sequence -- and what follows must be the next header or the end.
// Returns the index behind the record (the start of the next one); 0 if [i, n) does not hold a whole well-formed record
// at i.  at_eof: n is the end of the input (the last line may lack its newline).
static size_t record_end(const char *b, size_t i, size_t n, bool at_eof)
{
    if (i >= n) return 0;
    auto line_end = [&](size_t p) { const char *e = (const char *)std::memchr(b + p, '\n', n - p); return e ? (size_t)(e - b) : n; };
    auto graph_len = [&](size_t p, size_t e) { while (e > p && !std::isgraph((unsigned char)b[e - 1])) --e; return e - p; };
    if (b[i] == '>') {                                   // FASTA: up to the next '>' at a line start
        size_t p = line_end(i);
        if (p >= n) return at_eof ? n : 0;
        for (++p; p < n; ) { if (b[p] == '>') return p; const size_t e = line_end(p); if (e >= n) return at_eof ? n : 0; p = e + 1; }
        return at_eof ? n : 0;
    }
    if (b[i] != '@') return 0;
    size_t p = line_end(i);
    if (p >= n) return 0;
    ++p;
    size_t S = 0, Q = 0;
    for (;;) {                                           // sequence lines
        if (p >= n) return 0;
        if (b[p] == '+') break;
        const size_t e = line_end(p);
        if (e >= n) return 0;
        S += graph_len(p, e); p = e + 1;
    }
    if (S == 0) return 0;                                // no boundary is placed on an empty record: a quality line '@..' followed by
                                                         // one '+..' and one '@..' would verify as one
    { const size_t e = line_end(p); if (e >= n) return 0; p = e + 1; }       // the '+' line
    while (Q < S) {                                      // quality lines
        if (p >= n) return 0;
        const size_t e = line_end(p);
        Q += graph_len(p, e);
        if (e >= n) return (Q == S && at_eof) ? n : 0;
        p = e + 1;
    }
    if (Q != S) return 0;
    if (p < n && b[p] != '@') return 0;
    return p;
}
// first verified record start at or behind `from` (a line start is looked for first), looking at no more than max_lines
// lines; n if there is none.  One record can verify by coincidence when the candidate is a quality line (header and
// sequence of the next record counted as "sequence", lengths happening to add up), so three records in a row must verify
// -- or the input must end behind fewer (at_eof only: a window of a stream that ends earlier rejects the candidate).
static size_t find_record_start(const char *b, size_t from, size_t n, bool at_eof, int max_lines, char mark /* '@' FASTQ, '>' FASTA: the input's first byte */)
{
    size_t i = from;
    if (i > 0 && i < n && b[i - 1] != '\n') { const char *e = (const char *)std::memchr(b + i, '\n', n - i); i = e ? (size_t)(e - b) + 1 : n; }
    for (int t = 0; t < max_lines && i < n; ++t) {
        if (b[i] == mark) {                              // (a FASTQ quality line may start with '>' as well as with '@')
            size_t p = i; int good = 0;
            for (; good < 3 && p < n; ++good) { const size_t e = record_end(b, p, n, at_eof); if (!e) break; p = e; }
            if (good == 3 || (good > 0 && p >= n && at_eof)) return i;
        }
        const char *e = (const char *)std::memchr(b + i, '\n', n - i);
        i = e ? (size_t)(e - b) + 1 : n;
    }
    return n;
}
// record starts that split [lo, hi) of the file image into about `parts` pieces; a cut is made only where a record start
// verifies -- otherwise that piece simply stays larger
static std::vector<size_t> cut_records(const char *b, size_t lo, size_t hi, int parts)
{
    std::vector<size_t> cut(1, lo);
    if (parts > 1 && hi - lo > (size_t)(1 << 20) && (b[lo] == '@' || b[lo] == '>')) {
        for (int t = 1; t < parts; ++t) {
            const size_t i = find_record_start(b, lo + (hi - lo) / (size_t)parts * (size_t)t, hi, true, 64, b[lo]);
            if (i < hi && i > cut.back()) cut.push_back(i);
        }
    }
    cut.push_back(hi);
    return cut;
}
static double g_t_fread = 0, g_t_cut = 0, g_t_par = 0, g_t_merge = 0;   // PS_VERBOSE >= 2: where the parser's time goes
// parse [lo, hi) of the file image on `threads` threads and APPEND the reads to rs.  Every thread parses its range into arrays of its
// own; the ranges' sizes then give every range its place in rs, and the threads copy their parts there side by side (one thread
// joining the parts cost more than the parsing: 0.5 s against 0.3 s per 5 M reads on 8 cores).
static void parse_span(const char *b, size_t lo, size_t hi, int threads, ReadSet &rs)
{
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    const auto tc0 = std::chrono::steady_clock::now();
    const std::vector<size_t> cut = cut_records(b, lo, hi, threads);
    const int parts = (int)cut.size() - 1;
    const auto tc1 = std::chrono::steady_clock::now();
    std::vector<ReadSet> piece((size_t)parts);
    std::vector<char> anyq((size_t)parts, 0);
    auto work = [&](int t) {
        ReadSet &r = piece[t];
        const size_t bytes = cut[t + 1] - cut[t];
        r.seq.reserve(bytes / 2 + 64); r.qual.reserve(bytes / 2 + 64);          // a FASTQ record is at most half bases
        r.off.push_back(0); r.name_off.push_back(0);
        bool aq = false;
        parse_reads_range(b, cut[t], cut[t + 1], r, aq);
        anyq[t] = aq;
    };
    auto fan = [&](const std::function<void(int)> &f) { std::vector<std::thread> th; for (int t = 1; t < parts; ++t) th.emplace_back(f, t); f(0); for (auto &x : th) x.join(); };
    fan(work);
    const auto tc2 = std::chrono::steady_clock::now();
    g_t_cut += std::chrono::duration<double>(tc1 - tc0).count(); g_t_par += std::chrono::duration<double>(tc2 - tc1).count();
    if (rs.off.empty()) { rs.off.push_back(0); rs.name_off.push_back(0); }
    std::vector<size_t> n0((size_t)parts + 1), s0((size_t)parts + 1), m0((size_t)parts + 1);
    n0[0] = (size_t)rs.n; s0[0] = rs.seq.size(); m0[0] = rs.names.size();
    bool any_qual = rs.has_qual;
    for (int t = 0; t < parts; ++t) {
        n0[t + 1] = n0[t] + (size_t)piece[t].n; s0[t + 1] = s0[t] + piece[t].seq.size(); m0[t + 1] = m0[t] + piece[t].names.size();
        any_qual = any_qual || anyq[t];
    }
    const size_t tn = n0[parts], ts = s0[parts], tm = m0[parts];
    auto grow = [](auto &v, size_t need) { if (v.capacity() < need) v.reserve(need + need / 2); v.resize(need); };     // in large steps: a piece is appended to window by window
    grow(rs.len, tn); grow(rs.off, tn + 1); grow(rs.name_off, tn + 1); grow(rs.seq, ts); grow(rs.qual, ts); grow(rs.names, tm);
    auto place = [&](int t) {
        ReadSet &r = piece[t];
        if (r.n == 0) return;
        std::memcpy(rs.len.data() + n0[t], r.len.data(), (size_t)r.n * sizeof(int32_t));
        int64_t *o = rs.off.data() + n0[t], *m = rs.name_off.data() + n0[t];
        const int64_t so = (int64_t)s0[t], no = (int64_t)m0[t];
        for (int64_t k = 1; k <= r.n; ++k) { o[k] = r.off[k] + so; m[k] = r.name_off[k] + no; }
        std::memcpy(rs.seq.data() + s0[t], r.seq.data(), r.seq.size());
        std::memcpy(rs.qual.data() + s0[t], r.qual.data(), r.qual.size());
        std::memcpy(rs.names.data() + m0[t], r.names.data(), r.names.size());
        r = ReadSet();
    };
    fan(place);
    rs.n = (int64_t)tn; rs.has_qual = any_qual;
    g_t_merge += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc2).count();
}
static void parser_times(int threads)
{
    if (const char *e = std::getenv("PS_VERBOSE")) if (std::atoi(e) >= 2)
        std::fprintf(stderr, "[parasuite-hip]     parser: reading %.0f ms, cutting %.0f ms, parsing on %d threads %.0f ms, placing the threads' parts %.0f ms (sums over the windows)\n", 1e3 * g_t_fread, 1e3 * g_t_cut, threads, 1e3 * g_t_par, 1e3 * g_t_merge);
    g_t_fread = g_t_cut = g_t_par = g_t_merge = 0;
}
// `want` bytes at file offset `at` into dst, by a few threads side by side when the file is a regular one (one thread copies ~3 GB/s out
// of the page cache); returns the bytes read (fewer than wanted: the input ends there)
static size_t read_at(int fd, bool regular, off_t at, char *dst, size_t want, int threads)
{
    auto one = [&](size_t lo, size_t hi) -> size_t {
        size_t have = lo;
        while (have < hi) {
            const ssize_t r = regular ? ::pread(fd, dst + have, hi - have, at + (off_t)have) : ::read(fd, dst + have, hi - have);
            if (r < 0) { if (errno == EINTR) continue; throw Error("read error on the reads file"); }
            if (r == 0) break;
            have += (size_t)r;
        }
        return have - lo;
    };
    const int nt = regular ? (int)std::max<size_t>(1, std::min<size_t>((size_t)std::min(threads, 8), want >> 22)) : 1;      // >= 4 MB per thread
    if (nt == 1) return one(0, want);
    std::vector<size_t> got((size_t)nt, 0); std::vector<std::string> err((size_t)nt);
    auto part = [&](int t) { try { got[t] = one(want * (size_t)t / nt, want * (size_t)(t + 1) / nt); } catch (const std::exception &e) { err[t] = e.what(); } };
    { std::vector<std::thread> th; for (int t = 1; t < nt; ++t) th.emplace_back(part, t); part(0); for (auto &x : th) x.join(); }
    size_t total = 0;
    for (int t = 0; t < nt; ++t) {
        if (!err[t].empty()) throw Error(err[t]);
        total += got[t];
        if (got[t] < want * (size_t)(t + 1) / nt - want * (size_t)t / nt) break;       // the input ended inside this part: what lies behind is not there
    }
    return total;
}
struct FdCloser { int fd; ~FdCloser() { if (fd >= 0) ::close(fd); } };

void load_reads(const char *path, ReadSet &rs, int threads)
{
    rs = ReadSet();
    load_reads_chunked(path, threads, ~(size_t)0 >> 2, [&](ReadSet &&piece) { rs = std::move(piece); });
    if (rs.off.empty()) { rs.off.push_back(0); rs.name_off.push_back(0); }
}

void load_reads_chunked(const char *path, int threads, size_t chunk_bytes, const std::function<void(ReadSet &&)> &sink, size_t first_bytes,
                        const std::function<bool()> *hungry, size_t hungry_min_bytes)
{
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) throw Error(std::string("cannot open reads ") + path);
    FdCloser closer{fd};
    struct stat st;
    const bool regular = ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
    if (chunk_bytes < 4096) chunk_bytes = 4096;
    size_t unit = (size_t)64 << 20;
    if (const char *e = std::getenv("PS_UNIT_MB")) unit = (size_t)std::max(1, std::atoi(e)) << 20;
    unit = std::min(unit, chunk_bytes);
    // the first piece may be smaller (the stages behind the parser start sooner), the following ones double up to chunk_bytes
    size_t cur = first_bytes && first_bytes < chunk_bytes ? std::max<size_t>(first_bytes, 4096) : chunk_bytes;
    RawVec<char> buf; size_t have = 0; bool eof = false; char mark = 0; off_t file_at = 0;
    ReadSet acc; size_t acc_bytes = 0;
    auto flush = [&]() { if (acc.n) { sink(std::move(acc)); cur = std::min(chunk_bytes, cur * 2); } acc = ReadSet(); acc_bytes = 0; };
    while (!eof || have) {
        const size_t fine = std::min<size_t>((size_t)1 << 20, unit);                                  // a piece ends within this of its size
        const size_t win = std::min(unit, cur > acc_bytes + fine ? cur - acc_bytes : unit);          // the last window of a piece is what is missing to its size (a piece already at its size is taking the end of the input along: whole windows)
        const size_t want = win > have ? win : have + win;       // what is carried over from a window that could not be cut fills a window alone: it grows
        if (buf.size() < want + 1) buf.resize(want + 1);
        const auto tr0 = std::chrono::steady_clock::now();
        if (!eof && have < want) {
            const size_t got = read_at(fd, regular, file_at, buf.data() + have, want - have, threads);
            if (got < want - have) eof = true;
            have += got; file_at += (off_t)got;
        }
        g_t_fread += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
        if (!mark && have) mark = buf[0];
        size_t cut = have;
        if (!eof) {
            // the last record start that verifies: walk records from a start found in the window's last 256 KB
            const size_t from = have > ((size_t)256 << 10) ? have - ((size_t)256 << 10) : 0;
            size_t i = find_record_start(buf.data(), from, have, false, 4096, mark), last = 0;
            while (i < have) { last = i; const size_t e = record_end(buf.data(), i, have, false); if (!e || e >= have) break; i = e; }
            if (last == 0) continue;                                               // nothing to cut at: the window grows
            cut = last;
        }
        if (cut) {
            const size_t tiny = cur / 8;                                                                        // an end of the input not worth a launch of its own
            const bool to_the_end = regular && (size_t)std::max<off_t>(0, st.st_size - file_at) + have <= tiny;  // this window and all behind it
            if (acc_bytes && acc_bytes + cut > cur + fine && !to_the_end) flush();    // this window would take the piece well over its size (a window that had to grow)
            const bool first_window = acc.n == 0;
            parse_span(buf.data(), 0, cut, threads, acc);
            if (first_window && acc.n && cur > cut && cur < (~(size_t)0 >> 3)) {     // a piece's arrays are sized once, from what its first window held
                const double f = 1.05 * (double)cur / (double)cut;
                acc.len.reserve((size_t)(f * (double)acc.n) + 64); acc.off.reserve((size_t)(f * (double)acc.n) + 65); acc.name_off.reserve((size_t)(f * (double)acc.n) + 65);
                acc.seq.reserve((size_t)(f * (double)acc.seq.size()) + 64); acc.qual.reserve((size_t)(f * (double)acc.seq.size()) + 64);
                acc.names.reserve((size_t)(f * (double)acc.names.size()) + 64);
            }
            acc_bytes += cut;
            // no piece is cut off just in front of the end of the input: what is left would be a launch of its own (>= 0.3 s for 0.6 M reads,
            // measured) -- the piece takes it along, up to an eighth over its size
            const size_t rest = regular ? (size_t)std::max<off_t>(0, st.st_size - file_at) + (have - cut) : ~(size_t)0;
            const bool tiny_rest = !eof && rest <= tiny;
            if (!tiny_rest && (acc_bytes + fine > cur || (hungry && acc_bytes >= hungry_min_bytes && (*hungry)()))) flush();
        }
        std::memmove(buf.data(), buf.data() + cut, have - cut);
        have -= cut;
        if (eof && cut == 0) break;
    }
    flush();
    parser_times(threads);
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

// host loops over millions of reads: contiguous ranges on a few threads; f(begin, end, thread)
template <class F> static void par_for(size_t n, int threads, F f)
{
    size_t nt = (size_t)std::max(1, threads);
    if (nt > n / 8192 + 1) nt = n / 8192 + 1;
    if (nt <= 1) { if (n) f((size_t)0, n, 0); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(nt);
    const size_t per = (n + nt - 1) / nt;
    for (size_t t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
            const size_t a0 = t * per, b0 = std::min(n, a0 + per);
            try { if (a0 < b0) f(a0, b0, (int)t); } catch (...) { err[t] = std::current_exception(); }
        });
    for (auto &x : th) x.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
}
static int par_threads(size_t n, int threads) { size_t nt = (size_t)std::max(1, threads); if (nt > n / 8192 + 1) nt = n / 8192 + 1; return (int)std::max<size_t>(1, nt); }

// ------------------------------------------------------------ batch set-up ---
// host half of batch_create: bins, order inside the bins, 2-bit packing -- no device call, so the parser thread of
// ps_map runs it while the GPU thread is busy with the piece before
std::unique_ptr<Batch> batch_prepare(Ctx *ctx, ReadSet &&rs_in, int threads)
{
    const auto t_prep0 = std::chrono::steady_clock::now();
    struct PrepTimes { double bins = 0, sort = 0, pack = 0; } pt;
    std::unique_ptr<Batch> b(new Batch());
    b->ctx = ctx; b->rs = std::move(rs_in);
    const ReadSet &rs = b->rs;
    // cost class of a length: everything of the search model that depends on the length except the length itself
    std::vector<int> class_of_len;                       // length -> bin (-1: not seen yet)
    std::map<std::vector<int>, int> bin_of_class;
    b->read_bin.resize((size_t)rs.n); b->read_local.resize((size_t)rs.n);
    for (int64_t g = 0; g < rs.n; ++g) {
        const int len = rs.len[g];
        if (len < 1) throw Error("empty read in input");
        if ((size_t)len >= class_of_len.size()) class_of_len.resize((size_t)len + 64, -1);
        int cls = class_of_len[(size_t)len];
        if (cls < 0) {
            Model md; std::string err;
            if (!make_model(ctx->opt, len, md, err)) throw Error(err);
            // What a launch needs to be uniform in: the seed rule and the gap limit.  The difference budget, the number of score buckets,
            // the packed word count and where the N mask lives follow the read (budget: BtArgs::units_by_len) or the longest read of
            // the bin (local-memory layout), so that adapter-trimmed input with dozens of lengths is ONE launch where it used to be
            // one per budget step -- every launch ends with ~0.2 s of emptying machine
            const std::vector<int> key = {md.max_gapo};      // the seed rule follows the read too (len > seed_len, checked per read by the kernels) since adapter-trimmed
                                                             // PAR-CLIP reads lie on both sides of the 32-base seed: 18-40 bp input is one launch, not two
            auto bc = bin_of_class.find(key);
            if (bc == bin_of_class.end()) { bc = bin_of_class.emplace(key, (int)b->bins.size()).first; b->bins.emplace_back(); }
            cls = class_of_len[(size_t)len] = bc->second;
        }
        Bin &bin = b->bins[(size_t)cls];
        if (bin.len && bin.len != len) bin.ragged = true;
        if (len > bin.len) bin.len = len;
        b->read_bin[g] = cls; b->read_local[g] = (int32_t)bin.ids.size();
        bin.ids.push_back((int32_t)g);
    }
    pt.bins = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prep0).count();
    for (Bin &bin : b->bins) {
        std::string err;
        if (!make_model(ctx->opt, bin.len, bin.md, err)) throw Error(err);
        // Order the reads of a bin by their leading bases (the search consumes a read from its first base): the
        // lanes of a wave then walk the same top levels of the BWT, so their Occ loads coalesce and hit in cache.
        // Results return to input order through ids[]; the order inside a bin is free.  (Handing the kernels a sorted
        // LIST instead and leaving the reads where they are cost the search kernel 1.6 % and the width stage 20 %:
        // profiles/r03_kernel_experiments.txt.)  An LSD radix sort over the first 16 bases, every pass on all threads:
        // counts per thread and digit, places from their prefix sums (stable), 0.15 s per 3.6 M reads when one thread did it.
        if (!std::getenv("PS_KEEP_ORDER")) {         // PS_KEEP_ORDER=1: the reads stay in input order (tools/order_probe.py hands them out in an order of its own)
            const size_t n = bin.ids.size();
            const int nt = par_threads(n, threads);
            std::vector<uint32_t> key(n), key2(n); std::vector<int32_t> id2(n);
            par_for(n, nt, [&](size_t r0, size_t r1, int) {
                for (size_t r = r0; r < r1; ++r) {
                    const uint8_t *sq = rs.seq.data() + rs.off[bin.ids[r]];
                    const int kb = rs.len[bin.ids[r]] < 16 ? rs.len[bin.ids[r]] : 16;
                    uint32_t k = 0;
                    for (int j = 0; j < kb; ++j) k = (k << 2) | (uint32_t)(sq[j] & 3);
                    key[r] = k << (2 * (16 - kb));
                }
            });
            std::vector<size_t> cnt((size_t)nt * 256);
            for (int sh = 0; sh < 32; sh += 8) {
                std::fill(cnt.begin(), cnt.end(), 0);
                par_for(n, nt, [&](size_t r0, size_t r1, int t) { size_t *c = cnt.data() + (size_t)t * 256; for (size_t r = r0; r < r1; ++r) ++c[(key[r] >> sh) & 0xff]; });
                size_t at = 0;
                for (int d = 0; d < 256; ++d) for (int t = 0; t < nt; ++t) { const size_t c = cnt[(size_t)t * 256 + d]; cnt[(size_t)t * 256 + d] = at; at += c; }
                par_for(n, nt, [&](size_t r0, size_t r1, int t) {
                    size_t *c = cnt.data() + (size_t)t * 256;
                    for (size_t r = r0; r < r1; ++r) { const size_t p = c[(key[r] >> sh) & 0xff]++; key2[p] = key[r]; id2[p] = bin.ids[r]; }
                });
                key.swap(key2); bin.ids.swap(id2);
            }
            par_for(n, nt, [&](size_t r0, size_t r1, int) { for (size_t r = r0; r < r1; ++r) b->read_local[bin.ids[r]] = (int32_t)r; });
        }
        const size_t n = bin.ids.size();
        pt.sort = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prep0).count() - pt.bins - pt.pack;
        bin.n_bw = (bin.len + 15) / 16; bin.n_mw = (bin.len + 31) / 32;
        bin.h_bases.resize((size_t)bin.n_bw * n); bin.h_nmask.resize((size_t)bin.n_mw * n);
        bin.lens.resize(n);
        {
            const int nt = std::max(1, std::min(threads, 64));
            auto pack = [&](int t) {                                  // distinct reads write distinct words: no sharing.  Whole words, bases behind the read's end 0
                // the reads come in leading-base order, i.e. from all over the piece: three dependent cache misses per read (its offset,
                // its length, its bases) unless they are asked for ahead (257 ms per 9.3 M reads on 16 threads without)
                const size_t r_end = n * (t + 1) / nt, AHEAD = 12;
                for (size_t r = n * t / nt; r < r_end; ++r) {
                    if (r + 2 * AHEAD < r_end) { const int32_t g2 = bin.ids[r + 2 * AHEAD]; __builtin_prefetch(&rs.off[g2]); __builtin_prefetch(&rs.len[g2]); }
                    if (r + AHEAD < r_end) { const uint8_t *q = rs.seq.data() + rs.off[bin.ids[r + AHEAD]]; __builtin_prefetch(q); __builtin_prefetch(q + 63); }
                    const uint8_t *s = rs.seq.data() + rs.off[bin.ids[r]];
                    const int rl = rs.len[bin.ids[r]];
                    bin.lens[r] = rl;
                    for (int p = 0; p < bin.n_bw; ++p) {
                        uint32_t wd = 0;
                        const int j1 = std::min(rl, 16 * p + 16);
                        for (int j = 16 * p; j < j1; ++j) if (s[j] <= 3) wd |= (uint32_t)s[j] << (2 * (j & 15));
                        bin.h_bases[(size_t)p * n + r] = wd;
                    }
                    for (int p = 0; p < bin.n_mw; ++p) {
                        uint32_t wd = 0;
                        const int j1 = std::min(rl, 32 * p + 32);
                        for (int j = 32 * p; j < j1; ++j) if (s[j] > 3) wd |= 1u << (j & 31);
                        bin.h_nmask[(size_t)p * n + r] = wd;
                    }
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back(pack, t);
            pack(0);
            for (auto &x : th) x.join();
        }
        pt.pack = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prep0).count() - pt.bins - pt.sort;
    }
    if (const char *e = std::getenv("PS_VERBOSE")) if (std::atoi(e) >= 2)
        std::fprintf(stderr, "[parasuite-hip]     piece of %lld reads made ready in %.0f ms (bins %.0f, leading-base sort %.0f, packing %.0f)\n", (long long)rs.n,
                     1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prep0).count(), 1e3 * pt.bins, 1e3 * pt.sort, 1e3 * pt.pack);
    return b;
}
// device half: allocate and upload
void batch_upload(Batch &bb)
{
    Batch *b = &bb; Ctx *ctx = b->ctx;
    require_device(ctx->device);
    b->wk = b->work_index >= 0 ? ctx->work_at(b->work_index) : ctx->take_work();
    Work *wk = b->wk;
    for (Bin &bin : b->bins) {
        const size_t n = bin.ids.size();
        if (bin.ragged) { bin.d_lens.alloc(n); bin.d_lens.upload(bin.lens.data(), n, wk->stream); }
        bin.d_ids.alloc(n); bin.d_ids.upload(bin.ids.data(), n, wk->stream);
        bin.bases.alloc(bin.h_bases.size()); bin.nmask.alloc(bin.h_nmask.size());
        bin.bases.upload(bin.h_bases.data(), bin.h_bases.size(), wk->stream);
        bin.nmask.upload(bin.h_nmask.data(), bin.h_nmask.size(), wk->stream);
    }
    b->d_stats.alloc(3);
    PS_HIP(hipStreamSynchronize(wk->stream));
}
std::unique_ptr<Batch> batch_create(Ctx *ctx, ReadSet &&rs_in)
{
    std::unique_ptr<Batch> b = batch_prepare(ctx, std::move(rs_in), ctx->host_threads);
    batch_upload(*b);
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

// hand-out order of a search launch: queue position -> read, heaviest estimated search first, the given (leading-base) order inside a class
__global__ void k_order_keys(const uint8_t *est, int n, int cap, uint8_t *key, int32_t *iota)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const int e = est[r] > cap ? cap : est[r];
        key[r] = (uint8_t)(cap - e); iota[r] = r;
    }
}

__global__ void k_iota(int n, int32_t *iota) { for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) iota[r] = r; }

struct EvTimer {
    hipEvent_t a, b; hipStream_t s;
    explicit EvTimer(hipStream_t st) : s(st) { PS_HIP(hipEventCreate(&a)); PS_HIP(hipEventCreate(&b)); PS_HIP(hipEventRecord(a, s)); }
    double stop() { PS_HIP(hipEventRecord(b, s)); PS_HIP(hipEventSynchronize(b)); float ms = 0; PS_HIP(hipEventElapsedTime(&ms, a, b)); return ms; }
    // begin / end on the context's clock (ms since its reference event): launches of two streams that overlap in time
    void span(hipEvent_t ref, double &t_begin, double &t_end) { float x = 0, y = 0; if (ref && hipEventElapsedTime(&x, ref, a) == hipSuccess && hipEventElapsedTime(&y, ref, b) == hipSuccess) { t_begin = x; t_end = y; } }
    ~EvTimer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
};


// width + backtracking kernels over n reads of one length that are already packed on the device
static void run_search(Batch &b, const Model &md, int n, const uint32_t *d_bases, const uint32_t *d_nmask, const int32_t *d_lens,
                       uint32_t pool_cap, int aln_cap, AlnRec *alns, int32_t *n_aln, uint8_t *status, bool first_tier = false)
{
    Ctx *ctx = b.ctx; Work *wk = b.wk; hipStream_t s = wk->stream;
    const int len = md.len, seed_len = md.seed_len;
    uint32_t *w = wk->ws_get<uint32_t>("w", (size_t)(len + 1) * n);
    uint32_t *cwb = wk->ws_get<uint32_t>("cwb", (size_t)lm_ncw(len) * n);
    uint32_t *cswb = wk->ws_get<uint32_t>("cswb", (size_t)(lm_ncsw(seed_len) + 1) * n);
    WidthArgs wa;
    wa.ix = ctx->ix.view; wa.n_reads = n; wa.len = len; wa.lens = d_lens; wa.seed_len = seed_len; wa.use_seed = md.use_seed;
    wa.bases = d_bases; wa.nmask = d_nmask; wa.w = w; wa.cwb = cwb; wa.cswb = cswb; wa.stats = b.d_stats.p + 0;
    { EvTimer t(s); launch_width(wa, s); PS_HIP(hipGetLastError()); b.tm.ms_width += t.stop(); ++b.tm.n_width_launches; }
    // ---- hand-out order: the reads with the heaviest predicted search first (ps_effort.hip), so that the launch does not end on
    // them.  PS_ORDER=0 switches it off, 2 orders by the estimated best score alone (A/B runs).
    const int32_t *d_order = nullptr; const uint8_t *d_est = nullptr; const uint16_t *d_est_ab = nullptr;
    {
        const char *eo = std::getenv("PS_ORDER");
        const int mode = eo ? std::atoi(eo) : 1;
        int min_n = 4096;                                     // below that every read has a lane to itself at once: no order to choose
        if (const char *e = std::getenv("PS_ORDER_MIN")) min_n = std::max(1, std::atoi(e));      // tests: the small launches of the fuzz sweep too
        if (mode > 0 && n >= min_n && md.max_units >= md.c_min) {      // a search that can afford no difference is ~len steps for every read: nothing to order
            EvTimer t(s);
            uint8_t *est = wk->ws_get<uint8_t>("est", (size_t)n), *key = wk->ws_get<uint8_t>("okey", (size_t)n), *key2 = wk->ws_get<uint8_t>("okey2", (size_t)n);
            int32_t *iota = wk->ws_get<int32_t>("oiota", (size_t)n), *order = wk->ws_get<int32_t>("order", (size_t)n);
            EffortArgs ea;
            ea.ix = ctx->ix.view; ea.n_reads = n; ea.len = len; ea.lens = d_lens; ea.bases = d_bases; ea.nmask = d_nmask; ea.est = est;
            // everything here is in BUDGET UNITS (what the search's limits are in): the profile model has units == score, stock counts
            // every difference as one unit whatever it scores
            int csum = 0;
            for (int c = 0; c < 5; ++c) ea.s_pk[c] = md.u_mm_pk[c];
            for (int sc = 0; sc < 4; ++sc) for (int tc = 0; tc < 4; ++tc) if (sc != tc) csum += md.u_mm[sc][tc];
            ea.c_restart = std::max(1, (csum + 6) / 12);                      // an average mismatch
            if (const char *e = std::getenv("PS_ORDER_RESTART")) ea.c_restart = std::max(1, std::atoi(e));
            ea.w_pin = 16;
            if (const char *e = std::getenv("PS_ORDER_WPIN")) ea.w_pin = (uint32_t)std::max(1, std::atoi(e));
            int lv = 0; while (lv < 31 && (ctx->ix.view.seq_len >> (2 * lv)) > 0) ++lv;              // 4^lv > rows: 17 at hg19 size
            ea.est_ab = ctx->want_read_iters ? wk->ws_get<uint16_t>("est_ab", (size_t)n) : nullptr;
            launch_effort(ea, s);
            d_est_ab = ea.est_ab;
            int bits = 8;
            if (mode == 2) {
                int cap = 255;
                if (const char *e = std::getenv("PS_ORDER_CAP")) cap = std::max(1, std::min(255, std::atoi(e)));
                bits = 1; while ((1 << bits) <= cap) ++bits;
                hipLaunchKernelGGL(k_order_keys, dim3(std::min((n + 255) / 256, 4096)), dim3(256), 0, s, est, n, cap, key, iota);
            } else {
                EffortModelArgs em;
                em.n_reads = n; em.len = len; em.lens = d_lens; em.units_by_len = nullptr; em.bases = d_bases; em.nmask = d_nmask; em.cwb = cwb; em.est = est;
                for (int c = 0; c < 5; ++c) em.s_pk[c] = md.u_mm_pk[c];
                em.inv_c_min = (uint32_t)md.inv_c_min; em.max_units = md.max_units; em.u_tight = md.u_tight;
                em.seed_units = md.max_seed_diff * md.u_tight; em.use_seed = md.use_seed; em.seed_len = md.seed_len;
                em.max_gapo = md.max_gapo; em.indel_end_skip = md.indel_end_skip; em.u_gapo_ins = md.u_gapo_ins; em.u_gapo_del = md.u_gapo_del;
                em.depth = lv + 3; em.rows = (float)ctx->ix.view.seq_len; em.log_scale = 8;
                if (const char *e = std::getenv("PS_ORDER_SCALE")) em.log_scale = std::max(1, std::min(12, std::atoi(e)));
                if (d_lens) {                                             // ragged launch: every read's own budget, by its length
                    uint8_t *tab = wk->pin_get<uint8_t>("units_by_len_h2", 256);
                    for (int l2 = 0; l2 < 256; ++l2) { const int u = budget_diffs(ctx->opt, l2) * (ctx->opt.profile ? ctx->opt.unit : 1); tab[l2] = (uint8_t)(u > 255 ? 255 : u); }
                    uint8_t *d_tab = wk->ws_get<uint8_t>("units_by_len2", 256);
                    PS_HIP(hipMemcpyAsync(d_tab, tab, 256, hipMemcpyHostToDevice, s));
                    em.units_by_len = d_tab;
                }
                em.key = key; em.pred = nullptr;
                launch_effort_model(em, s);
                hipLaunchKernelGGL(k_iota, dim3(std::min((n + 255) / 256, 4096)), dim3(256), 0, s, n, iota);
            }
            size_t tb = 0;
            PS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key, key2, iota, order, n, 0, bits, s));
            uint8_t *tmp = wk->ws_get<uint8_t>("order_tmp", tb ? tb : 1);
            PS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, key, key2, iota, order, n, 0, bits, s));
            PS_HIP(hipGetLastError());
            b.tm.ms_width += t.stop();                                        // reported with the width stage: both prepare the search
            d_order = order; d_est = est;
        }
    }
    int dev_cus = 256;
    { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, ctx->device) == hipSuccess && p.multiProcessorCount > 0) dev_cus = p.multiProcessorCount; }
    // narrow entries link with 16-bit indices, keep one 64-bit bucket bitmap and count inserted / deleted bases in 3 bits each
    // ... and saturate the count of best hits at 255 (it is only ever compared with max_top2)
    const bool wide = pool_cap > 65535 || md.n_buckets > 64 || md.max_gapo + md.max_gape > 7 || md.max_top2 >= 255;
    const int lm = lm_bytes(len, seed_len, md.n_buckets, wide);
    int per_cu = (int)((size_t)(160 * 1024) / ((size_t)256 * lm));
    if (per_cu < 1) throw Error("read length / score range too large for the per-lane LDS state");
    { const char *e = std::getenv("PS_MAX_PER_CU"); const int cap = e ? std::atoi(e) : 4; if (per_cu > cap) per_cu = cap; }   // workgroups per CU the kernel's registers allow (ps_kernels.hip: PS_BT_WAVES)
    int blocks = ctx->bt_blocks > 0 ? ctx->bt_blocks : dev_cus * per_cu;
    int need = (n + 255) / 256;
    if (blocks > need) blocks = need;
    // bound the lanes by stack memory (the widest tier keeps 64 MB per lane)
    const size_t per_lane = (size_t)pool_cap * (wide ? sizeof(Entry) : 16) + (wide ? PS_MAX_BUCKETS * 4 : 0);
    const size_t max_lanes = ((size_t)(wide ? 32 : 64) << 30) / per_lane;
    if ((size_t)blocks * 256 > max_lanes) blocks = (int)std::max<size_t>(1, max_lanes / 256);
    const int n_lanes = blocks * 256;
    uint8_t *pool = wk->ws_get<uint8_t>("pool", (size_t)n_lanes * pool_cap * (wide ? sizeof(Entry) : 16));
    uint32_t *heads = wide ? wk->ws_get<uint32_t>("heads", (size_t)n_lanes * PS_MAX_BUCKETS) : nullptr;
    uint32_t *queue = wk->ws_get<uint32_t>("queue", 16);
    PS_HIP(hipMemsetAsync(queue, 0, 64, s));
    BtArgs a; std::memset(&a, 0, sizeof a);
    a.ix = ctx->ix.view; a.md = md; a.n_reads = n; a.len = len; a.lens = d_lens; a.n_lanes = n_lanes;
    if (d_lens) {                                               // ragged launch: every read's own budget, by its length
        uint8_t *tab = wk->pin_get<uint8_t>("units_by_len_h", 256);
        for (int l2 = 0; l2 < 256; ++l2) { const int u = budget_diffs(ctx->opt, l2) * (ctx->opt.profile ? ctx->opt.unit : 1); tab[l2] = (uint8_t)(u > 255 ? 255 : u); }
        uint8_t *d_tab = wk->ws_get<uint8_t>("units_by_len", 256);
        PS_HIP(hipMemcpyAsync(d_tab, tab, 256, hipMemcpyHostToDevice, s));
        a.units_by_len = d_tab;
    }
    a.bases = d_bases; a.nmask = d_nmask; a.n_bw = (len + 15) / 16; a.n_mw = (len + 31) / 32;
    a.w = w; a.cwb = cwb; a.cswb = cswb;
    a.alns = alns; a.aln_cap = aln_cap; a.n_aln = n_aln; a.status = status;
    a.pool = pool; a.pool_cap = pool_cap; a.heads = heads; a.wide = wide ? 1 : 0; a.stats = b.d_stats.p + 1;
    a.queue = queue; a.fetch_min = ctx->fetch_min; a.hit_min = ctx->hit_min;
    a.order = wide ? nullptr : d_order; a.est = d_est; a.est_ab = d_est_ab;
    // the estimate also spares the search entries (ps_narrow.h, nt_tail): first tier and profile costs only (units == score); a read it fails on starts over without it inside the launch
    a.cap_est = (first_tier && !wide && d_est && md.profile && !(std::getenv("PS_CAP") && std::atoi(std::getenv("PS_CAP")) == 0)) ? 1 : 0;
    if (a.cap_est) if (const char *e = std::getenv("PS_CAP_BIAS")) a.cap_est += std::max(0, std::min(200, std::atoi(e)));      // tests: estimates too low by that much, so that the restart path runs
    if (const char *e = std::getenv("PS_FETCH_MIN")) a.fetch_min = std::max(1, std::atoi(e));       // tuning: read at every launch
    if (const char *e = std::getenv("PS_HIT_MIN")) a.hit_min = std::max(1, std::atoi(e));
    if (!wide && pool_cap < 65535 && ctx->n_big > 0) {         // large slots for the reads that outgrow their private slice
        a.big_cap = 65535; a.n_big = (uint32_t)std::min<int64_t>(ctx->n_big, std::max(64, n));
        a.big_pool = wk->ws_get<uint8_t>("big_pool", (size_t)a.n_big * a.big_cap * 16);
        a.big_next = queue + 4;                                  // second counter in the zeroed queue words
        a.big_busy = wk->ws_get<uint32_t>("big_busy", a.n_big);
        PS_HIP(hipMemsetAsync(a.big_busy, 0, (size_t)a.n_big * 4, s));
    }
    uint32_t *riters = nullptr;
    if (ctx->want_read_iters) { riters = wk->ws_get<uint32_t>("riters", (size_t)n * PS_RI_WORDS); PS_HIP(hipMemsetAsync(riters, 0, (size_t)n * PS_RI_WORDS * 4, s)); a.read_iters = riters; }
    { EvTimer t(s);
      if (!launch_backtrack(a, wk->ws_get<BtArgs>("btargs", 1), wk->pin_get<BtArgs>("btargs_h", 1), blocks, lm, s, ctx->want_kstats || ctx->want_read_iters)) throw Error("cost model outside the ranges the search kernel packs (gap/score fields must fit a byte)");
      PS_HIP(hipGetLastError());
      const double ms = t.stop(); b.tm.ms_backtrack += ms; ++b.tm.n_backtrack_launches;
      { double t0_ = 0, t1_ = 0; t.span(ctx->ref_event, t0_, t1_); if (b.tm.n_backtrack_launches == 1) b.tm.bt_begin_ms = t0_; b.tm.bt_end_ms = t1_; }
      if (std::getenv("PS_VERBOSE")) std::fprintf(stderr, "[parasuite-hip]   backtrack launch: %d reads x %d bp, stack %u%s, %d lanes, %.1f ms\n", n, len, pool_cap, wide ? " (wide)" : "", n_lanes, ms); }
    if (ctx->want_read_iters) { ctx->read_iters.resize((size_t)n * PS_RI_WORDS); PS_HIP(hipMemcpyAsync(ctx->read_iters.data(), riters, (size_t)n * PS_RI_WORDS * 4, hipMemcpyDeviceToHost, s)); PS_HIP(hipStreamSynchronize(s)); }
}

// ------------------------------------------------------------- host helpers ------
// download the hit lists of n reads (stride aln_cap on the device) in compact form
static void download_alns(Work *wk, int n, int aln_cap, const AlnRec *d_alns, const int32_t *d_n_aln,
                          std::vector<int32_t> &n_aln, std::vector<uint32_t> &off, std::vector<AlnRec> &alns)
{
    hipStream_t s = wk->stream;
    uint32_t *cnt = wk->ws_get<uint32_t>("cnt", n), *d_off = wk->ws_get<uint32_t>("off", (size_t)n + 1);
    hipLaunchKernelGGL(k_clip_counts, dim3((n + 255) / 256), dim3(256), 0, s, d_n_aln, aln_cap, n, cnt);
    size_t tb = 0;
    PS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt, d_off, n, s));
    uint8_t *tmp = wk->ws_get<uint8_t>("scan_tmp", tb ? tb : 1);
    PS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cnt, d_off, n, s));
    int32_t *p_na = wk->pin_get<int32_t>("dl_n_aln", n); uint32_t *p_off = wk->pin_get<uint32_t>("dl_off", (size_t)n + 1);
    PS_HIP(hipMemcpyAsync(p_na, d_n_aln, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    PS_HIP(hipMemcpyAsync(p_off, d_off, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    PS_HIP(hipStreamSynchronize(s));
    uint32_t last = 0;
    if (n) { int m = p_na[n - 1]; last = p_off[n - 1] + (uint32_t)(m > aln_cap ? aln_cap : (m < 0 ? 0 : m)); }
    p_off[n] = last;
    n_aln.assign(p_na, p_na + n); off.assign(p_off, p_off + n + 1); alns.resize(last);
    if (last) {
        AlnRec *comp = wk->ws_get<AlnRec>("comp", last);
        hipLaunchKernelGGL(k_gather_alns, dim3((n + 255) / 256), dim3(256), 0, s, d_alns, aln_cap, d_n_aln, d_off, n, comp);
        AlnRec *p_al = wk->pin_get<AlnRec>("dl_alns", last);
        PS_HIP(hipMemcpyAsync(p_al, comp, (size_t)last * sizeof(AlnRec), hipMemcpyDeviceToHost, s));
        PS_HIP(hipStreamSynchronize(s));
        std::memcpy(alns.data(), p_al, (size_t)last * sizeof(AlnRec));
    }
}

// ------------------------------------------------- samse stage on the device --------
// Read classes for the tie-break stream (one drand48 stream over all reads in input order):
//   0 no hit (no draw) | 1 exactly one best-score SA interval (always two draws) | 2 several (data dependent)
// bit 2 (PS_CLS_HOST): the read is finished on the host -- class 2 (sequential chain), reads that list
// alternative hits (XA), reads that needed a larger search tier.  Everything else never leaves the GPU.
static const uint8_t PS_CLS_HOST = 4;

__global__ void k_classify(const AlnRec *alns, int aln_cap, const int32_t *n_aln, const uint8_t *status, const int32_t *ids,
                           int n, int n_occ, uint8_t *cls_out)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        uint8_t c = 0;
        if (status[r] != RS_OK) c = 3 | PS_CLS_HOST;        // hit list lives on the host (larger tier): class fixed there
        else {
            const int na = n_aln[r];
            if (na > 0) {
                const AlnRec *al = alns + (size_t)r * aln_cap;
                const int best = al[0].score;
                int nb = 0; unsigned long long tot = 0;
                for (int j = 0; j < na; ++j) { if (al[j].score == best && nb == j) ++nb; tot += (unsigned long long)(al[j].l - al[j].k) + 1ull; }
                c = nb == 1 ? 1 : 2;
                if (c == 2 || (n_occ > 0 && tot >= 2 && tot <= (unsigned long long)n_occ + 1ull)) c |= PS_CLS_HOST;
            }
        }
        cls_out[ids[r]] = c;
    }
}
__global__ void k_class_flags(const uint8_t *cls, long long n, uint32_t *is1, uint32_t *is2)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const int c = cls[g] & 3;
        is1[g] = c == 1; is2[g] = c == 2;
    }
}
__global__ void k_gather_sub(const AlnRec *alns, int aln_cap, const int32_t *n_aln, const int32_t *local, int m, AlnRec *out, int32_t *n_out)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < m; q += gridDim.x * blockDim.x) {
        const int r = local[q]; int na = n_aln[r]; if (na > aln_cap) na = aln_cap;
        n_out[q] = na;
        for (int j = 0; j < na; ++j) out[(size_t)q * aln_cap + j] = alns[(size_t)r * aln_cap + j];
    }
}

__device__ __forceinline__ unsigned long long lcg_jump(unsigned long long x, unsigned long long t)
{
    const unsigned long long M = 0xFFFFFFFFFFFFULL;
    unsigned long long a = 0x5DEECE66DULL, c = 0xBULL, ra = 1, rc = 0;
    while (t) {
        if (t & 1) { ra = (ra * a) & M; rc = (rc * a + c) & M; }
        c = (c * a + c) & M; a = (a * a) & M;
        t >>= 1;
    }
    return (ra * x + rc) & M;
}

struct SelectArgs {
    const AlnRec *alns; int aln_cap; const int32_t *n_aln; const int32_t *ids; int n;
    const uint8_t *cls; const uint32_t *e_before; const uint32_t *h_before; const unsigned long long *hard_cum;
    unsigned long long draws_in;
    SelRec *sel; bwtint *rows; int *err;
};
// ---- the choice of a read's main hit (upstream bwa_aln2seq_core), ONE copy for the device kernel and the host-finished reads ----
// Walks the best-score intervals in list order: every one costs a draw, the one that wins costs a second draw that places the hit
// inside it.  x is the state of the drand48 stream (advanced by the draws made, whose number is returned); c1 / c2 = occurrences
// at the best score / at the other listed scores.  IEEE doubles in this order of operations on both sides.
struct MainPick { bwtint sa; int32_t c1, c2; int type, n_mm, n_gapo, n_gape, ref_shift, score; };
__host__ __device__ inline unsigned long long lcg48_next(unsigned long long x) { return (x * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL; }
__host__ __device__ inline int rule_choose_main(const AlnRec *al, int na, unsigned long long &x, MainPick &h)
{
    int cnt = 0, draws = 0, i;
    const int best = al[0].score;
    h.sa = 0; h.n_mm = h.n_gapo = h.n_gape = h.ref_shift = h.score = 0;
    for (i = 0; i < na; ++i) {
        const AlnRec p = al[i];
        if (p.score > best) break;
        const unsigned long long wdt = (unsigned long long)(p.l - p.k) + 1ull;
        x = lcg48_next(x); ++draws;
        if ((double)x * (1.0 / 281474976710656.0) * (double)(wdt + (unsigned long long)(long long)cnt) > (double)cnt) {
            h.n_mm = p.n_mm; h.n_gapo = p.n_gapo; h.n_gape = p.n_gape;
            h.ref_shift = (int)p.n_del - (int)p.n_ins; h.score = p.score;
            x = lcg48_next(x); ++draws;
            h.sa = p.k + (bwtint)((double)wdt * ((double)x * (1.0 / 281474976710656.0)));
        }
        cnt += (int)wdt;
    }
    h.c1 = cnt;
    for (; i < na; ++i) cnt += (int)((unsigned long long)(al[i].l - al[i].k) + 1ull);
    h.c2 = cnt - h.c1;
    h.type = h.c1 > 1 ? 2 : 1;
    return draws;
}

// the single-best reads: two draws at a stream position known from the prefix counts
__global__ void k_select(SelectArgs a)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < a.n; r += gridDim.x * blockDim.x) {
        const int g = a.ids[r];
        const uint8_t c = a.cls[g];
        SelRec s; s.sa = 0; s.c1 = s.c2 = 0; s.type = 0; s.n_mm = s.n_gapo = s.n_gape = 0; s.ref_shift = 0; s.score = 0; s.pad[0] = s.pad[1] = 0;
        if (c == 1) {                       // class 1, finished on the device
            const AlnRec *al = a.alns + (size_t)r * a.aln_cap;
            const int na = a.n_aln[r];
            const unsigned int hb = a.h_before[g];
            const unsigned long long off = a.draws_in + 2ull * a.e_before[g] + (hb ? a.hard_cum[hb - 1] : 0ull);
            unsigned long long x = lcg_jump((11ull << 16) | 0x330Eull, off);
            if (lcg48_next(x) == 0) *a.err = 1;       // the offsets assume two draws per such read: the first draw wins unless it is exactly 0
            MainPick pk;
            (void)rule_choose_main(al, na, x, pk);
            s.sa = pk.sa; s.c1 = pk.c1; s.c2 = pk.c2; s.type = (uint8_t)pk.type;
            s.n_mm = (uint8_t)pk.n_mm; s.n_gapo = (uint8_t)pk.n_gapo; s.n_gape = (uint8_t)pk.n_gape; s.ref_shift = (int8_t)pk.ref_shift; s.score = (uint8_t)pk.score;
        }
        a.sel[g] = s;
        a.rows[g] = s.type ? s.sa : 0;
    }
}

struct PostArgs {
    const int32_t *ids; int n, len; const int32_t *lens; long long l_pac;
    const uint8_t *cls; const SelRec *sel; const bwtint *pos; FinRec *fin;
    int budget, profile, unit; const uint8_t *logn;     // MAPQ rule inputs; logn[n] = (int)(4.343 ln n + .5)
    const uint8_t *budget_by_len;                        // with lens: the difference budget of a read of every length (budget: the longest read's)
    RefineItem *items; int32_t *item_g; unsigned int *n_items;
};
// ---- the two samse rules every finished hit goes through, ONE copy for the device kernel and for the host-finished subset ----
// text position of an SA row -> forward coordinate of the alignment's first base and its strand (upstream bwa_sa2pos /
// bwa_cal_pac_pos_core); -1: the alignment spans the forward/reverse junction
__host__ __device__ inline long long rule_to_forward(long long pos_t, long long l_pac, int ref_len, int &strand)
{
    long long pos_f = pos_t;
    strand = 0;
    if (pos_f < l_pac && l_pac < pos_f + ref_len) return -1;
    const bool is_rev = pos_f >= l_pac;
    if (is_rev) pos_f = 2 * l_pac - 1 - pos_f;
    strand = !is_rev;
    if (is_rev) pos_f = pos_f + 1 < ref_len ? 0 : pos_f - ref_len + 1;
    return pos_f;
}
// upstream bwa_approx_mapQ with the budget rule of the cost model in use; logn[n] = (int)(4.343 ln n + .5), n < 256
__host__ __device__ inline int rule_mapq(int c1, int c2, int n_mm, int score, int budget, bool profile, int unit, const uint8_t *logn)
{
    if (c1 == 0) return 23;
    if (c1 > 1) return 0;
    if (!profile) { if (n_mm == budget) return 25; }
    else if (budget * unit - score < unit) return 25;
    if (c2 == 0) return 37;
    const int lg = logn[c2 >= 255 ? 255 : c2];
    return 23 < lg ? 0 : 23 - lg;
}
static void mapq_logn_table(uint8_t logn[256])
{
    logn[0] = 0;
    for (int n = 1; n < 256; ++n) logn[n] = (uint8_t)(int)(4.343 * std::log((double)n) + 0.5);
}

// text position -> forward coordinate + strand, MAPQ; gapped hits are queued for the banded-DP kernel
__global__ void k_post(PostArgs a)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < a.n; r += gridDim.x * blockDim.x) {
        const int g = a.ids[r];
        FinRec f; f.pos = -1; f.strand = 0; f.mapq = 0; f.type = 0; f.pad[0] = f.pad[1] = f.pad[2] = f.pad[3] = f.pad[4] = 0;
        const SelRec s = a.sel[g];
        if (a.cls[g] == 1 && s.type != 0) {
            const int ref_len = (a.lens ? a.lens[r] : a.len) + s.ref_shift;
            int strand = 0;
            const long long p = rule_to_forward((long long)a.pos[g], a.l_pac, ref_len, strand);
            const int budget = (a.lens && a.budget_by_len) ? (int)a.budget_by_len[a.lens[r]] : a.budget;
            const int mq = rule_mapq(s.c1, s.c2, s.n_mm, (int)s.score, budget, a.profile != 0, a.unit, a.logn);
            f.pos = p; f.strand = (uint8_t)strand; f.mapq = (uint8_t)mq; f.type = p < 0 ? 0 : s.type;
            if (f.type != 0 && s.n_gapo) {
                const unsigned int q = atomicAdd(a.n_items, 1u);
                a.items[q] = RefineItem{r, (bwtint)p, (int32_t)s.ref_shift, strand};
                a.item_g[q] = g;
            }
        }
        a.fin[g] = f;
    }
}

// The big allocations of a lane of work (tier-1 stack slices, the large stack slots) made ahead of its first search: a
// hipMalloc of tens of GB synchronises the device, so when a second worker makes its own while the first worker's search kernel
// runs it waits for that kernel (seen as a 1-2 s stall of a piece).  ps_map calls this when a worker starts, before any search.
void reserve_search_workspace(Ctx *ctx, int work_index)
{
    require_device(ctx->device);
    Work *wk = ctx->work_at(work_index);
    int dev_cus = 256;
    { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, ctx->device) == hipSuccess && p.multiProcessorCount > 0) dev_cus = p.multiProcessorCount; }
    const uint32_t pool_cap = ctx->pool_cap[0];
    if (pool_cap > 65535) return;                             // first tier is the wide one: sized by the launch
    int blocks = ctx->bt_blocks > 0 ? ctx->bt_blocks : dev_cus * 4;
    const size_t per_lane = (size_t)pool_cap * 16, max_lanes = ((size_t)64 << 30) / per_lane;
    if ((size_t)blocks * 256 > max_lanes) blocks = (int)std::max<size_t>(1, max_lanes / 256);
    (void)wk->ws_get<uint8_t>("pool", (size_t)blocks * 256 * per_lane);
    if (pool_cap < 65535 && ctx->n_big > 0) {
        (void)wk->ws_get<uint8_t>("big_pool", (size_t)ctx->n_big * 65535 * 16);
        (void)wk->ws_get<uint32_t>("big_busy", (size_t)ctx->n_big);
    }
}

// --------------------------------------------------------------- search stage -------
void batch_search(Batch &b)
{
    Ctx *ctx = b.ctx; Work *wk = b.wk; hipStream_t s = wk->stream;
    require_device(ctx->device);
    b.tm = Timing();
    auto t0 = Clock::now();
    const int64_t N = b.rs.n;
    b.d_stats.zero(s);
    for (int t = 0; t < 3; ++t) b.n_overflow[t] = 0;
    if (b.d_class.n < (size_t)N) b.d_class.alloc((size_t)N);
    for (Bin &bin : b.bins) {
        const int n = (int)bin.ids.size();
        bin.host_alns_valid = false;
        const int cap1 = (bin.len <= 40 && ctx->aln_cap_short > ctx->aln_cap[0]) ? ctx->aln_cap_short : ctx->aln_cap[0];
        if (bin.d_alns.n < (size_t)n * cap1 || bin.aln_cap != cap1) { bin.d_alns.alloc((size_t)n * cap1); bin.d_n_aln.alloc(n); bin.d_status.alloc(n); }
        bin.aln_cap = cap1;
        run_search(b, bin.md, n, bin.bases.p, bin.nmask.p, bin.ragged ? bin.d_lens.p : nullptr, ctx->pool_cap[0], bin.aln_cap, bin.d_alns.p, bin.d_n_aln.p, bin.d_status.p, true);
        uint8_t *h_status = wk->pin_get<uint8_t>("status", n);
        PS_HIP(hipMemcpyAsync(h_status, bin.d_status.p, (size_t)n, hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(k_classify, dim3(std::min((n + 255) / 256, 4096)), dim3(256), 0, s, bin.d_alns.p, bin.aln_cap, bin.d_n_aln.p, bin.d_status.p,
                           bin.d_ids.p, n, ctx->opt.n_occ, b.d_class.p);
        PS_HIP(hipStreamSynchronize(s));
        bin.overflow.clear();
        std::vector<int32_t> todo;
        for (int r = 0; r < n; ++r) {
            if (h_status[r] == RS_OK) continue;
            if (h_status[r] == RS_BAD_SCORE) throw Error("internal: score outside the bucket range");
            todo.push_back(r);
        }
        // reads that need a deeper stack / a longer hit list: the second narrow tier, then the wide one.  A read whose stack outgrew
        // 65,535 entries already (RS_OVERFLOW_DEEP: on a large slot of the first launch) skips the second tier, which has no more than
        // that: on a repeat-rich genome those are the longest searches of the batch, and every tier starts them from scratch
        std::vector<int32_t> deep;
        { std::vector<int32_t> keep; for (int32_t r : todo) (h_status[r] == RS_OVERFLOW_DEEP ? deep : keep).push_back(r); todo.swap(keep); }
        for (int tier = 1; tier < 3; ++tier) {
            if (tier == 2) { todo.insert(todo.end(), deep.begin(), deep.end()); std::sort(todo.begin(), todo.end()); deep.clear(); }
            if (todo.empty()) continue;
            b.n_overflow[tier] += (int64_t)todo.size();
            const int m = (int)todo.size();
            std::vector<uint32_t> hb((size_t)bin.n_bw * m), hm((size_t)bin.n_mw * m);
            for (int q = 0; q < m; ++q) {
                for (int wv = 0; wv < bin.n_bw; ++wv) hb[(size_t)wv * m + q] = bin.h_bases[(size_t)wv * n + todo[q]];
                for (int wv = 0; wv < bin.n_mw; ++wv) hm[(size_t)wv * m + q] = bin.h_nmask[(size_t)wv * n + todo[q]];
            }
            DevBuf<uint32_t> db, dm; db.alloc(hb.size()); dm.alloc(hm.size());
            db.upload(hb.data(), hb.size(), s); dm.upload(hm.data(), hm.size(), s);
            std::vector<int32_t> hl(m); DevBuf<int32_t> dl;
            if (bin.ragged) { for (int q = 0; q < m; ++q) hl[q] = bin.lens[todo[q]]; dl.alloc(m); dl.upload(hl.data(), m, s); }
            DevBuf<AlnRec> ta; DevBuf<int32_t> tn; DevBuf<uint8_t> ts;
            ta.alloc((size_t)m * ctx->aln_cap[tier]); tn.alloc(m); ts.alloc(m);
            run_search(b, bin.md, m, db.p, dm.p, bin.ragged ? dl.p : nullptr, ctx->pool_cap[tier], ctx->aln_cap[tier], ta.p, tn.p, ts.p);
            std::vector<uint8_t> st(m); ts.download(st.data(), m, s);
            std::vector<int32_t> na; std::vector<uint32_t> off; std::vector<AlnRec> al;
            download_alns(wk, m, ctx->aln_cap[tier], ta.p, tn.p, na, off, al);
            std::vector<int32_t> still;
            for (int q = 0; q < m; ++q) {
                if (st[q] == RS_BAD_SCORE) throw Error("internal: score outside the bucket range");
                if (st[q] != RS_OK) { still.push_back(todo[q]); continue; }
                bin.overflow[todo[q]] = std::vector<AlnRec>(al.begin() + off[q], al.begin() + off[q + 1]);
            }
            todo.swap(still);
        }
        if (!todo.empty()) throw Error("a read exceeded the largest search tier (stack or hit capacity)");
    }
    KStats hs[3];
    b.d_stats.download(hs, 3, s);
    // classes to the host; the host-finished subset and its position in the tie-break stream
    auto tcl = Clock::now();
    b.h_class = (uint8_t *)b.p_class.get((size_t)N + 64);
    PS_HIP(hipMemcpyAsync(b.h_class, b.d_class.p, (size_t)N, hipMemcpyDeviceToHost, s));
    PS_HIP(hipStreamSynchronize(s));
    b.st_width = hs[0]; b.st_backtrack = hs[1];
    bool patched = false;
    for (Bin &bin : b.bins)
        for (auto &kv : bin.overflow) {                       // larger-tier reads: class from their host-side hit list
            const std::vector<AlnRec> &al = kv.second;
            int nb = 0;
            for (; nb < (int)al.size() && al[nb].score == al[0].score; ++nb) {}
            b.h_class[bin.ids[kv.first]] = (uint8_t)((al.empty() ? 0 : (nb == 1 ? 1 : 2)) | PS_CLS_HOST);
            patched = true;
        }
    if (patched) PS_HIP(hipMemcpyAsync(b.d_class.p, b.h_class, (size_t)N, hipMemcpyHostToDevice, s));
    b.sub.clear(); b.n_class1 = 0; b.n_hard = 0;
    {
        // pass 1: per range, the number of class-1 / class-2 reads and of reads the host finishes
        const int nt = par_threads((size_t)N, ctx->host_threads);
        std::vector<int64_t> ce(nt + 1, 0), ch(nt + 1, 0), cs(nt + 1, 0);
        par_for((size_t)N, ctx->host_threads, [&](size_t g0, size_t g1, int t) {
            int64_t e = 0, h = 0, sn = 0;
            for (size_t g = g0; g < g1; ++g) { const uint8_t c = b.h_class[g]; e += (c & 3) == 1; h += (c & 3) == 2; sn += (c & PS_CLS_HOST) != 0; }
            ce[t + 1] = e; ch[t + 1] = h; cs[t + 1] = sn;
        });
        for (int t = 0; t < nt; ++t) { ce[t + 1] += ce[t]; ch[t + 1] += ch[t]; cs[t + 1] += cs[t]; }
        b.n_class1 = ce[nt]; b.n_hard = ch[nt];
        b.sub.resize((size_t)cs[nt]);
        // pass 2: the subset with its position in the tie-break stream
        par_for((size_t)N, ctx->host_threads, [&](size_t g0, size_t g1, int t) {
            int64_t e = ce[t], h = ch[t]; size_t q = (size_t)cs[t];
            for (size_t g = g0; g < g1; ++g) {
                const uint8_t c = b.h_class[g];
                if (c & PS_CLS_HOST) { SubRead &sr = b.sub[q++]; sr = SubRead(); sr.g = (int64_t)g; sr.cls = c & 3; sr.easy_before = e; sr.hard_before = h; }
                e += (c & 3) == 1; h += (c & 3) == 2;
            }
        });
        // hit lists of the subset: gathered on the device in subset order, one pinned download per bin
        std::vector<std::vector<int32_t>> want(b.bins.size());
        std::vector<int32_t> slot(b.sub.size(), -1);
        for (size_t q = 0; q < b.sub.size(); ++q) {
            const SubRead &sr = b.sub[q];
            const int bi = b.read_bin[sr.g]; const Bin &bin = b.bins[bi];
            if (bin.overflow.empty() || !bin.overflow.count(b.read_local[sr.g])) { slot[q] = (int32_t)want[bi].size(); want[bi].push_back(b.read_local[sr.g]); }
        }
        b.sub_alns.resize(b.bins.size());
        std::vector<const AlnRec *> got(b.bins.size(), nullptr); std::vector<const int32_t *> got_n(b.bins.size(), nullptr);
        for (size_t bi = 0; bi < b.bins.size(); ++bi) {
            const int m = (int)want[bi].size();
            if (!m) continue;
            Bin &bin = b.bins[bi];
            int32_t *d_loc = wk->ws_get<int32_t>("sub_local", m); AlnRec *d_out = wk->ws_get<AlnRec>("sub_alns", (size_t)m * bin.aln_cap);
            int32_t *d_no = wk->ws_get<int32_t>("sub_n", m);
            PS_HIP(hipMemcpyAsync(d_loc, want[bi].data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_gather_sub, dim3((m + 255) / 256), dim3(256), 0, s, bin.d_alns.p, bin.aln_cap, bin.d_n_aln.p, d_loc, m, d_out, d_no);
            if (!b.sub_alns[bi]) b.sub_alns[bi].reset(new PinBuf());
            const size_t bytes_al = (size_t)m * bin.aln_cap * sizeof(AlnRec);
            uint8_t *hp = (uint8_t *)b.sub_alns[bi]->get(bytes_al + (size_t)m * 4 + 64);
            PS_HIP(hipMemcpyAsync(hp, d_out, bytes_al, hipMemcpyDeviceToHost, s));
            PS_HIP(hipMemcpyAsync(hp + bytes_al, d_no, (size_t)m * 4, hipMemcpyDeviceToHost, s));
            got[bi] = reinterpret_cast<const AlnRec *>(hp); got_n[bi] = reinterpret_cast<const int32_t *>(hp + bytes_al);
        }
        PS_HIP(hipStreamSynchronize(s));
        par_for(b.sub.size(), ctx->host_threads, [&](size_t q0, size_t q1, int) {
            for (size_t q = q0; q < q1; ++q) {
                SubRead &sr = b.sub[q];
                const int bi = b.read_bin[sr.g]; Bin &bin = b.bins[bi];
                if (slot[q] < 0) { const std::vector<AlnRec> &v = bin.overflow.find(b.read_local[sr.g])->second; sr.alns = v.data(); sr.n_alns = (int32_t)v.size(); }
                else { sr.alns = got[bi] + (size_t)slot[q] * bin.aln_cap; sr.n_alns = got_n[bi][slot[q]]; }
            }
        });
    }
    b.tm.ms_classify = ms_since(tcl);
    b.searched = true; b.selected_hard = b.selected = b.located = false;
    b.tm.ms_total = ms_since(t0);
}

void Batch::ensure_host_alns()
{
    require_device(ctx->device);
    for (Bin &bin : bins) {
        if (bin.host_alns_valid) continue;
        download_alns(wk, (int)bin.ids.size(), bin.aln_cap, bin.d_alns.p, bin.d_n_aln.p, bin.h_n_aln, bin.h_off, bin.h_alns);
        for (auto &kv : bin.overflow) bin.h_n_aln[kv.first] = (int32_t)kv.second.size();
        bin.host_alns_valid = true;
    }
}
const AlnRec *Batch::alns_of(int64_t g, int &n)
{
    ensure_host_alns();
    const Bin &bin = bins[read_bin[g]];
    int32_t r = read_local[g];
    n = bin.h_n_aln[r];
    if (!bin.overflow.empty()) {
        auto it = bin.overflow.find(r);
        if (it != bin.overflow.end()) return it->second.data();
    }
    return bin.h_alns.data() + bin.h_off[r];
}

// ----------------------------------------------------- tie-break selection -----
// Among the hits with the best score one occurrence is chosen at random; the reference's aligner
// draws from ONE drand48 stream (seed 11) over all reads in input order.
static int choose_main(const AlnRec *al, int na, Rng48 &rng, Hit &h)
{
    MainPick pk;
    unsigned long long x = rng.x;
    const int draws = rule_choose_main(al, na, x, pk);
    rng.x = x;
    h.sa = pk.sa; h.c1 = pk.c1; h.c2 = pk.c2; h.type = pk.type;
    h.n_mm = pk.n_mm; h.n_gapo = pk.n_gapo; h.n_gape = pk.n_gape; h.ref_shift = pk.ref_shift; h.score = pk.score;
    return draws;
}

// the sequential part of the stream: reads with several best-score intervals, in input order
void batch_select_hard(Batch &b, uint64_t draws_before, uint64_t *draws_after)
{
    if (!b.searched) throw Error("select before search");
    auto t0 = Clock::now();
    b.draws_in = draws_before;
    b.hard_draws_cum.assign((size_t)b.n_hard, 0);
    Rng48 rng(11);
    rng.jump(draws_before);
    uint64_t H = 0; int64_t e_prev = 0;
    for (SubRead &sr : b.sub) {
        if (sr.cls != 2) continue;
        rng.jump(2ull * (uint64_t)(sr.easy_before - e_prev));      // the single-best reads in between took two draws each
        e_prev = sr.easy_before;
        sr.hit = Hit();
        H += (uint64_t)choose_main(sr.alns, sr.n_alns, rng, sr.hit);
        b.hard_draws_cum[(size_t)sr.hard_before] = H;
    }
    b.draws_out = draws_before + 2ull * (uint64_t)b.n_class1 + H;
    if (draws_after) *draws_after = b.draws_out;
    b.selected_hard = true;
    b.tm.ms_select += ms_since(t0); b.tm.ms_sel_hard = ms_since(t0);
}

void batch_select_easy(Batch &b, int threads)
{
    if (!b.selected_hard) throw Error("select_easy before select_hard");
    Ctx *ctx = b.ctx; Work *wk = b.wk; hipStream_t s = wk->stream;
    require_device(ctx->device);
    auto t0 = Clock::now();
    const int64_t N = b.rs.n;
    (void)threads;
    // ---- device: prefix counts of the two draw classes, then every single-best read picks its occurrence ----
    if (b.d_sel.n < (size_t)N) { b.d_sel.alloc((size_t)N); b.d_fin.alloc((size_t)N); b.d_rows.alloc((size_t)N + 1); b.d_pos.alloc((size_t)N + 1); b.d_eb.alloc((size_t)N); b.d_hb.alloc((size_t)N); }
    const double ms_alloc = ms_since(t0);
    {
        uint32_t *f1 = wk->ws_get<uint32_t>("flag1", (size_t)N), *f2 = wk->ws_get<uint32_t>("flag2", (size_t)N);
        hipLaunchKernelGGL(k_class_flags, dim3(2048), dim3(256), 0, s, b.d_class.p, (long long)N, f1, f2);
        size_t tb = 0;
        PS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, f1, b.d_eb.p, (size_t)N, s));
        uint8_t *tmp = wk->ws_get<uint8_t>("scan_tmp", tb ? tb : 1);
        PS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, f1, b.d_eb.p, (size_t)N, s));
        PS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, f2, b.d_hb.p, (size_t)N, s));
    }
    unsigned long long *d_cum = wk->ws_get<unsigned long long>("hard_cum", (size_t)b.n_hard + 1);
    if (b.n_hard) PS_HIP(hipMemcpyAsync(d_cum, b.hard_draws_cum.data(), (size_t)b.n_hard * 8, hipMemcpyHostToDevice, s));
    int *d_err = wk->ws_get<int>("sel_err", 4);
    PS_HIP(hipMemsetAsync(d_err, 0, 16, s));
    for (Bin &bin : b.bins) {
        SelectArgs a;
        a.alns = bin.d_alns.p; a.aln_cap = bin.aln_cap; a.n_aln = bin.d_n_aln.p; a.ids = bin.d_ids.p; a.n = (int)bin.ids.size();
        a.cls = b.d_class.p; a.e_before = b.d_eb.p; a.h_before = b.d_hb.p; a.hard_cum = d_cum; a.draws_in = b.draws_in;
        a.sel = b.d_sel.p; a.rows = b.d_rows.p; a.err = d_err;
        hipLaunchKernelGGL(k_select, dim3(std::min((a.n + 255) / 256, 4096)), dim3(256), 0, s, a);
    }
    const double ms_launch = ms_since(t0);
    // ---- host: the subset (its class-1 members by the same offset algebra, then the alternative-hit lists) ----
    const int n_occ = ctx->opt.n_occ;
    b.multis.clear();
    {
        const int nt = par_threads(b.sub.size(), threads);
        std::vector<std::vector<Multi>> part(nt);
        std::vector<size_t> first(nt + 1, 0);
        par_for(b.sub.size(), threads, [&](size_t q0, size_t q1, int t) {
            std::vector<Multi> &mine = part[t];
            first[t] = q0;
            for (size_t q = q0; q < q1; ++q) {
                SubRead &sr = b.sub[q];
                const int na = sr.n_alns;
                Hit &h = sr.hit;
                if (sr.cls == 0 || na == 0) { h = Hit(); h.type = 0; h.pos = -1; continue; }
                if (sr.cls == 1) {
                    h = Hit();
                    Rng48 rng(11);
                    rng.jump(b.draws_in + 2ull * (uint64_t)sr.easy_before + (sr.hard_before ? b.hard_draws_cum[(size_t)sr.hard_before - 1] : 0ull));
                    Rng48 probe = rng;
                    if (probe.step() == 0) throw Error("tie-break stream hit the zero state; sequential replay required");
                    choose_main(sr.alns, na, rng, h);
                }
                h.pos = -1; h.multi_begin = (int32_t)mine.size(); h.n_multi = 0;     // local index: shifted below
                if (n_occ > 0) {                     // alternative hits (samse -n): only if all occurrences of all hits number <= n_occ+1
                    int tot = 0;
                    for (int k = 0; k < na; ++k) tot += (int)((uint64_t)(sr.alns[k].l - sr.alns[k].k) + 1ull);
                    if (tot >= 0 && tot <= n_occ + 1)
                        for (int k = 0; k < na; ++k)
                            for (uint64_t row = sr.alns[k].k; row <= sr.alns[k].l; ++row) {
                                Multi m; std::memset(&m, 0, sizeof m);
                                m.row = (bwtint)row; m.gap = sr.alns[k].n_gapo + sr.alns[k].n_gape; m.mm = sr.alns[k].n_mm;
                                m.ref_shift = (int)sr.alns[k].n_del - (int)sr.alns[k].n_ins; m.pos = -1;
                                mine.push_back(m); ++h.n_multi;
                            }
                }
            }
        });
        std::vector<size_t> base(nt + 1, 0);
        for (int t = 0; t < nt; ++t) base[t + 1] = base[t] + part[t].size();
        b.multis.resize(base[nt]);
        const size_t per = b.sub.empty() ? 1 : (b.sub.size() + (size_t)nt - 1) / (size_t)nt;
        par_for((size_t)nt, nt, [&](size_t t0, size_t t1, int) {
            for (size_t t = t0; t < t1; ++t) {
                if (!part[t].empty()) std::memcpy(b.multis.data() + base[t], part[t].data(), part[t].size() * sizeof(Multi));
                const size_t q0 = t * per, q1 = std::min(b.sub.size(), q0 + per);
                for (size_t q = q0; q < q1; ++q) if (b.sub[q].cls != 0 && b.sub[q].n_alns != 0) b.sub[q].hit.multi_begin += (int32_t)base[t];
            }
        });
    }
    const double ms_host = ms_since(t0);
    int err = 0;
    PS_HIP(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, s));
    PS_HIP(hipStreamSynchronize(s));
    if (err) throw Error("tie-break stream hit the zero state; sequential replay required");
    if (const char *e = std::getenv("PS_VERBOSE")) if (std::atoi(e) >= 3)
        std::fprintf(stderr, "[parasuite-hip]     select_easy of %lld reads: allocations %.1f ms, launches until %.1f, host part until %.1f, device done at %.1f ms\n", (long long)N, ms_alloc, ms_launch, ms_host, ms_since(t0));
    b.selected = true;
    b.tm.ms_select += ms_since(t0); b.tm.ms_sel_easy = ms_since(t0);
}

// ------------------------------------------------- locate / MAPQ / gapped DP ----
static int fix_cigar(uint32_t *cigar, int n, int64_t &rb)
{
    if (n <= 0) return 0;
    if ((cigar[n - 1] & 0xf) == 1) cigar[n - 1] = (cigar[n - 1] >> 4 << 4) | 3;   // trailing insertion -> soft clip
    if ((cigar[0] & 0xf) == 1) cigar[0] = (cigar[0] >> 4 << 4) | 3;
    if ((cigar[n - 1] & 0xf) == 2) --n;                                            // trailing deletion dropped
    if (n > 0 && (cigar[0] & 0xf) == 2) { rb += cigar[0] >> 4; --n; std::memmove(cigar, cigar + 1, (size_t)n * 4); }
    return n;
}

// banded DP kernel over a list of items of one length bin; cigars come back to the host
static void run_refine(Batch &b, Bin &bin, const RefineItem *d_items, int n_it, std::vector<uint32_t> &cig, std::vector<int32_t> &nc)
{
    Ctx *ctx = b.ctx; Work *wk = b.wk; hipStream_t s = wk->stream;
    uint32_t *d_cig = wk->ws_get<uint32_t>("rf_cig", (size_t)n_it * PS_MAX_CIGAR); int32_t *d_nc = wk->ws_get<int32_t>("rf_nc", n_it);
    int blocks = (n_it + 63) / 64; if (blocks > 2048) blocks = 2048;
    const int tmax = bin.len + 64;
    RefineArgs ra;
    ra.ix = ctx->ix.view; ra.n_items = n_it; ra.len = bin.len; ra.lens = bin.ragged ? bin.d_lens.p : nullptr; ra.n_reads = (int)bin.ids.size();
    ra.bases = bin.bases.p; ra.nmask = bin.nmask.p; ra.items = d_items; ra.cigar = d_cig; ra.n_cigar = d_nc;
    ra.z_per_block = (size_t)64 * tmax * (bin.len < 2 * tmax + 1 ? bin.len : 2 * tmax + 1);
    ra.zbuf = wk->ws_get<uint8_t>("rf_z", ra.z_per_block * blocks);
    { EvTimer t(s); launch_refine(ra, blocks, s); PS_HIP(hipGetLastError()); b.tm.ms_refine += t.stop(); }
    cig.resize((size_t)n_it * PS_MAX_CIGAR); nc.resize(n_it);
    PS_HIP(hipMemcpyAsync(cig.data(), d_cig, cig.size() * 4, hipMemcpyDeviceToHost, s));
    PS_HIP(hipMemcpyAsync(nc.data(), d_nc, (size_t)n_it * 4, hipMemcpyDeviceToHost, s));
    PS_HIP(hipStreamSynchronize(s));
}

void batch_locate(Batch &b)
{
    if (!b.selected) throw Error("locate before select");
    Ctx *ctx = b.ctx; Work *wk = b.wk; hipStream_t s = wk->stream;
    require_device(ctx->device);
    const int64_t N = b.rs.n, l_pac = ctx->ix.ref.l_pac;
    auto t0 = Clock::now();
    // ---- device-finished reads: SA walk, strand / MAPQ, queue of gapped hits ----
    { EvTimer t(s); launch_sa2pos(ctx->ix.view, b.d_rows.p, b.d_pos.p, (int)N, b.d_stats.p + 2, s); PS_HIP(hipGetLastError()); b.tm.ms_sa2pos += t.stop(); }
    uint8_t logn[256];
    mapq_logn_table(logn);
    uint8_t *d_logn = wk->ws_get<uint8_t>("logn", 256);
    PS_HIP(hipMemcpyAsync(d_logn, logn, 256, hipMemcpyHostToDevice, s));
    uint8_t *h_budget = wk->pin_get<uint8_t>("budget_by_len_h", 256);
    for (int l2 = 0; l2 < 256; ++l2) { const int bd = budget_diffs(ctx->opt, l2); h_budget[l2] = (uint8_t)(bd > 255 ? 255 : bd); }
    uint8_t *d_budget = wk->ws_get<uint8_t>("budget_by_len", 256);
    PS_HIP(hipMemcpyAsync(d_budget, h_budget, 256, hipMemcpyHostToDevice, s));
    b.dev_cigars.clear();
    struct BinItems { RefineItem *d_items; int32_t *d_item_g; unsigned int *d_n; unsigned int n; };
    std::vector<BinItems> bi_items(b.bins.size());
    for (size_t bi = 0; bi < b.bins.size(); ++bi) {
        Bin &bin = b.bins[bi];
        const int n = (int)bin.ids.size();
        BinItems &it = bi_items[bi];
        it.d_items = wk->ws_get<RefineItem>("post_items" + std::to_string(bi), n); it.d_item_g = wk->ws_get<int32_t>("post_item_g" + std::to_string(bi), n);
        it.d_n = wk->ws_get<unsigned int>("post_n" + std::to_string(bi), 4);
        PS_HIP(hipMemsetAsync(it.d_n, 0, 16, s));
        PostArgs a;
        a.ids = bin.d_ids.p; a.n = n; a.len = bin.len; a.lens = bin.ragged ? bin.d_lens.p : nullptr; a.l_pac = l_pac; a.cls = b.d_class.p; a.sel = b.d_sel.p; a.pos = b.d_pos.p; a.fin = b.d_fin.p;
        a.budget = budget_diffs(ctx->opt, bin.len); a.profile = ctx->opt.profile; a.unit = ctx->opt.unit; a.logn = d_logn; a.budget_by_len = d_budget;
        a.items = it.d_items; a.item_g = it.d_item_g; a.n_items = it.d_n;
        hipLaunchKernelGGL(k_post, dim3(std::min((n + 255) / 256, 4096)), dim3(256), 0, s, a);
        PS_HIP(hipMemcpyAsync(&it.n, it.d_n, 4, hipMemcpyDeviceToHost, s));
    }
    b.h_sel = (SelRec *)b.p_sel.get((size_t)N * sizeof(SelRec) + 64);
    b.h_fin = (FinRec *)b.p_fin.get((size_t)N * sizeof(FinRec) + 64);
    PS_HIP(hipMemcpyAsync(b.h_sel, b.d_sel.p, (size_t)N * sizeof(SelRec), hipMemcpyDeviceToHost, s));
    PS_HIP(hipMemcpyAsync(b.h_fin, b.d_fin.p, (size_t)N * sizeof(FinRec), hipMemcpyDeviceToHost, s));
    PS_HIP(hipStreamSynchronize(s));
    PS_HIP(hipMemcpy(&b.st_sa2pos, b.d_stats.p + 2, sizeof(KStats), hipMemcpyDeviceToHost));
    for (size_t bi = 0; bi < b.bins.size(); ++bi) {            // gapped device-finished hits: banded DP, CIGAR clean-up
        BinItems &it = bi_items[bi];
        if (!it.n) continue;
        std::vector<uint32_t> cig; std::vector<int32_t> nc; std::vector<int32_t> gs(it.n);
        run_refine(b, b.bins[bi], it.d_items, (int)it.n, cig, nc);
        PS_HIP(hipMemcpy(gs.data(), it.d_item_g, (size_t)it.n * 4, hipMemcpyDeviceToHost));
        for (unsigned int q = 0; q < it.n; ++q) {
            const int64_t g = gs[q];
            uint32_t *c = cig.data() + (size_t)q * PS_MAX_CIGAR;
            int64_t rb = b.h_fin[g].pos;
            const int n_c = fix_cigar(c, nc[q], rb);
            if (n_c > PS_HIT_CIGAR) throw Error("CIGAR with more than 8 operations (raise PS_HIT_CIGAR for max_gapo > 2)");
            DevCigar dc; dc.g = g; dc.n = n_c; std::memcpy(dc.c, c, sizeof dc.c);
            b.dev_cigars.push_back(dc);
            b.h_fin[g].pos = rb;
            if (n_c == 0) b.h_fin[g].type = 0;
        }
    }
    std::sort(b.dev_cigars.begin(), b.dev_cigars.end(), [](const DevCigar &x, const DevCigar &y) { return x.g < y.g; });
    auto t1 = Clock::now();
    // ---- host-finished subset: its rows (main + alternatives) through the same SA kernel, then strand / MAPQ / DP ----
    const size_t M = b.sub.size(), n_rows = M + b.multis.size();
    if (n_rows) {
        std::vector<bwtint> rows(n_rows), pos(n_rows);
        par_for(M, ctx->host_threads, [&](size_t q0, size_t q1, int) { for (size_t q = q0; q < q1; ++q) rows[q] = b.sub[q].hit.type != 0 ? b.sub[q].hit.sa : 0; });
        par_for(b.multis.size(), ctx->host_threads, [&](size_t j0, size_t j1, int) { for (size_t j = j0; j < j1; ++j) rows[M + j] = b.multis[j].row; });
        bwtint *d_r = wk->ws_get<bwtint>("sub_rows", n_rows), *d_p = wk->ws_get<bwtint>("sub_pos", n_rows);
        PS_HIP(hipMemcpyAsync(d_r, rows.data(), n_rows * sizeof(bwtint), hipMemcpyHostToDevice, s));
        { EvTimer t(s); launch_sa2pos(ctx->ix.view, d_r, d_p, (int)n_rows, nullptr, s); PS_HIP(hipGetLastError()); b.tm.ms_sa2pos += t.stop(); }
        PS_HIP(hipMemcpyAsync(pos.data(), d_p, n_rows * sizeof(bwtint), hipMemcpyDeviceToHost, s));
        PS_HIP(hipStreamSynchronize(s));
        std::vector<std::vector<RefineItem>> items(b.bins.size());
        struct Back { size_t q; int32_t multi; };             // multi < 0: main hit
        std::vector<std::vector<Back>> back(b.bins.size());
        {
            const int nt = par_threads(M, ctx->host_threads);
            std::vector<std::vector<std::vector<RefineItem>>> t_items(nt, std::vector<std::vector<RefineItem>>(b.bins.size()));
            std::vector<std::vector<std::vector<Back>>> t_back(nt, std::vector<std::vector<Back>>(b.bins.size()));
            par_for(M, ctx->host_threads, [&](size_t q0, size_t q1, int t) {
                for (size_t q = q0; q < q1; ++q) {
                    SubRead &sr = b.sub[q]; Hit &h = sr.hit;
                    const int len = b.rs.len[sr.g], bi = b.read_bin[sr.g];
                    if (h.type != 0) {
                        int strand = 0;
                        h.pos = rule_to_forward((long long)pos[q], l_pac, len + h.ref_shift, strand);
                        h.strand = strand;
                        h.mapq = rule_mapq(h.c1, h.c2, h.n_mm, h.score, budget_diffs(ctx->opt, len), ctx->opt.profile, ctx->opt.unit, logn);
                        if (h.pos < 0) h.type = 0;
                    }
                    int kept = 0;
                    for (int j = 0; j < h.n_multi; ++j) {
                        Multi &m = b.multis[h.multi_begin + j];
                        int strand = 0;
                        m.pos = rule_to_forward((long long)pos[M + h.multi_begin + j], l_pac, len + m.ref_shift, strand);
                        m.strand = strand;
                        if (m.pos != h.pos && m.pos >= 0) b.multis[h.multi_begin + kept++] = m;
                    }
                    h.n_multi = kept;
                    for (int j = 0; j < h.n_multi; ++j) {
                        Multi &m = b.multis[h.multi_begin + j];
                        if (m.gap) { t_items[t][bi].push_back(RefineItem{b.read_local[sr.g], (bwtint)m.pos, m.ref_shift, m.strand}); t_back[t][bi].push_back(Back{q, j}); }
                    }
                    if (h.type != 0 && h.n_gapo) { t_items[t][bi].push_back(RefineItem{b.read_local[sr.g], (bwtint)h.pos, h.ref_shift, h.strand}); t_back[t][bi].push_back(Back{q, -1}); }
                }
            });
            for (int t = 0; t < nt; ++t)
                for (size_t bi = 0; bi < b.bins.size(); ++bi) {
                    items[bi].insert(items[bi].end(), t_items[t][bi].begin(), t_items[t][bi].end());
                    back[bi].insert(back[bi].end(), t_back[t][bi].begin(), t_back[t][bi].end());
                }
        }
        for (size_t bi = 0; bi < b.bins.size(); ++bi) {
            const int n_it = (int)items[bi].size();
            if (!n_it) continue;
            RefineItem *d_it = wk->ws_get<RefineItem>("sub_items", n_it);
            PS_HIP(hipMemcpyAsync(d_it, items[bi].data(), (size_t)n_it * sizeof(RefineItem), hipMemcpyHostToDevice, s));
            std::vector<uint32_t> cig; std::vector<int32_t> nc;
            run_refine(b, b.bins[bi], d_it, n_it, cig, nc);
            for (int q = 0; q < n_it; ++q) {
                const Back &bk = back[bi][q];
                Hit &h = b.sub[bk.q].hit;
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
        par_for(M, ctx->host_threads, [&](size_t q0, size_t q1, int) {          // alternatives whose gapped refinement produced nothing are dropped
            for (size_t q = q0; q < q1; ++q) {
                Hit &h = b.sub[q].hit;
                int kept = 0;
                for (int j = 0; j < h.n_multi; ++j) {
                    Multi &m = b.multis[h.multi_begin + j];
                    if (m.gap && m.n_cigar == 0) continue;
                    b.multis[h.multi_begin + kept++] = m;
                }
                h.n_multi = kept;
            }
        });
    }
    b.tm.ms_host_post += ms_since(t1);
    b.located = true;
    b.tm.ms_total += ms_since(t0);
}

// one read's alignment record: from the host-finished subset, else assembled from the device records
void Batch::hit_of(int64_t g, Hit &h) const
{
    if (h_class[g] & PS_CLS_HOST) {
        auto it = std::lower_bound(sub.begin(), sub.end(), g, [](const SubRead &s, int64_t v) { return s.g < v; });
        h = it->hit;
        return;
    }
    const SelRec &s = h_sel[g]; const FinRec &f = h_fin[g];
    h = Hit();
    h.sa = s.sa; h.type = f.type; h.pos = f.type ? f.pos : -1; h.strand = f.strand; h.mapq = f.mapq;
    h.n_mm = s.n_mm; h.n_gapo = s.n_gapo; h.n_gape = s.n_gape; h.ref_shift = s.ref_shift; h.score = s.score; h.c1 = s.c1; h.c2 = s.c2;
    if (s.n_gapo && s.type) {
        auto it = std::lower_bound(dev_cigars.begin(), dev_cigars.end(), g, [](const DevCigar &c, int64_t v) { return c.g < v; });
        if (it != dev_cigars.end() && it->g == g) { h.n_cigar = it->n; std::memcpy(h.cigar, it->c, sizeof h.cigar); }
    }
}

// ------------------------------------------------------------------ SAM -------
static inline int host_pac(const uint8_t *pac, int64_t p) { return (pac[(size_t)p >> 2] >> ((~p & 3) << 1)) & 3; }
static void put_int(std::string &o, long v)            // a dozen numbers per SAM line: no snprintf
{
    char b[24]; int n = 24;
    unsigned long u = v < 0 ? 0ul - (unsigned long)v : (unsigned long)v;
    do { b[--n] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) b[--n] = '-';
    o.append(b + n, (size_t)(24 - n));
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
    const uint8_t *pac = ref.pac_data();
    auto cmp = [&](int l) {
        for (int z = 0; z < l && x + z < ref.l_pac; ++z) {
            int c = host_pac(pac, x + z);
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
                for (int z = 0; z < l && x + z < ref.l_pac; ++z) md.push_back("ACGT"[host_pac(pac, x + z)]);
                u = 0; x += l; nm += l;
            }
        }
    } else cmp(len);
    put_int(md, u);
}

// SAM text goes through a raw cursor into storage the caller has made large enough (Room): a line is ~35 small pieces, and one
// std::string append per piece was most of the 1.3 us per read and thread that the writer -- the last stage of ps_map -- spent.
namespace {
struct Cur {
    char *p;
    inline void ch(char c) { *p++ = c; }
    inline void mem(const char *s, size_t n) { std::memcpy(p, s, n); p += n; }
    template <size_t N> inline void lit(const char (&s)[N]) { std::memcpy(p, s, N - 1); p += N - 1; }
    inline void num(long v)
    {
        static const char D2[] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
        unsigned long u = (unsigned long)v;
        if (v < 0) { *p++ = '-'; u = 0ul - u; }
        if (u < 10) { *p++ = (char)('0' + u); return; }
        if (u < 100) { std::memcpy(p, D2 + 2 * u, 2); p += 2; return; }
        char b[24]; int n = 24;
        while (u >= 100) { const unsigned long r = u % 100; u /= 100; n -= 2; std::memcpy(b + n, D2 + 2 * r, 2); }
        if (u >= 10) { n -= 2; std::memcpy(b + n, D2 + 2 * u, 2); } else b[--n] = (char)('0' + u);
        std::memcpy(p, b + n, (size_t)(24 - n)); p += 24 - n;
    }
    inline void cigar(int n, const uint32_t *c, int len)
    {
        if (n) for (int j = 0; j < n; ++j) { num((long)(c[j] >> 4)); ch("MIDS"[c[j] & 0xf]); }
        else { num(len); ch('M'); }
    }
};
// storage with `used` bytes taken: at least `need` more, the string's size being the storage (grown in large steps, never shrunk here)
inline char *room(std::string &o, size_t used, size_t need)
{
    if (o.size() < used + need) o.resize(std::max(o.size() + o.size() / 2, used + need + ((size_t)1 << 16)));
    return &o[0] + used;
}
}
// the XA list of a read (alternative hits, `samse -n 3`): chr,(+|-)pos,CIGAR,NM;
static void xa_cur(const Batch &b, const Hit &h, int len, std::string &o, size_t &used)
{
    const RefSeq &ref = b.ctx->ix.ref;
    for (int j = 0; j < h.n_multi; ++j) {
        const Multi &m = b.multis[h.multi_begin + j];
        int sid = 0;
        ref.cnt_ambi(m.pos, (int)ref_span(m.n_cigar, m.cigar, len), &sid);
        const Contig &mc = ref.contigs[sid];
        Cur c{room(o, used, mc.name.size() + 64 + 12 * (size_t)PS_MAX_CIGAR)};
        char *const c0 = c.p;
        c.mem(mc.name.data(), mc.name.size()); c.ch(','); c.ch(m.strand ? '-' : '+'); c.num((long)(m.pos - mc.offset + 1)); c.ch(',');
        c.cigar(m.n_cigar, m.cigar, len);
        c.ch(','); c.num(m.gap + m.mm); c.ch(';');
        used += (size_t)(c.p - c0);
    }
}
static void xa_text(const Batch &b, const Hit &h, int len, std::string &o)        // appended to a string (the BAM route)
{
    size_t used = o.size();
    xa_cur(b, h, len, o, used);
    o.resize(used);
}

// one line at o[used...]; `used` moves on.  o.size() is storage, not content (room()).
static void sam_line(const Batch &b, int64_t g, std::string &o, size_t &used)
{
    const ReadSet &rs = b.rs; const RefSeq &ref = b.ctx->ix.ref; const Options &opt = b.ctx->opt;
    Hit h; b.hit_of(g, h);
    const int len = rs.len[g];
    const uint8_t *seq = rs.seq.data() + rs.off[g];
    const char *qual = rs.has_qual ? rs.qual.data() + rs.off[g] : nullptr;
    size_t nl; const char *nm_ = rs.name(g, nl);
    int seqid = 0, nn = 0, span = 0;
    const Contig *ct = nullptr;
    if (h.type != 0) {
        span = (int)ref_span(h.n_cigar, h.cigar, len);
        nn = ref.cnt_ambi(h.pos, span, &seqid);
        ct = &ref.contigs[seqid];
    }
    // oriented read for MD/NM
    static thread_local std::string md; int nm = 0;
    if (h.type != 0) {
        uint8_t tmp_small[256]; std::vector<uint8_t> tmp_big;
        uint8_t *tmp = tmp_small;
        if (len > 256) { tmp_big.resize((size_t)len); tmp = tmp_big.data(); }
        const uint8_t *oriented = seq;
        if (h.strand) { for (int i = 0; i < len; ++i) { uint8_t c = seq[len - 1 - i]; tmp[i] = c > 3 ? c : (uint8_t)(3 - c); } oriented = tmp; }
        md.clear();
        cal_md(ref, h.n_cigar, h.cigar, len, h.pos, oriented, md, nm);
    }
    // everything but the XA list: name, 11 columns (two of them the read), at most 9 tags of <= 26 characters, MD
    Cur c{room(o, used, nl + 2 * (size_t)len + (ct ? ct->name.size() + md.size() : 0) + 12 * (size_t)PS_MAX_CIGAR + 384)};
    char *const c0 = c.p;
    c.mem(nm_, nl);
    auto put_seq = [&](int strand) {
        char *d = c.p;
        if (!strand) for (int i = 0; i < len; ++i) d[i] = "ACGTN"[seq[i]];
        else for (int i = 0; i < len; ++i) d[i] = "TGCAN"[seq[len - 1 - i]];
        d[len] = '\t';
        d += len + 1;
        if (qual) { if (!strand) std::memcpy(d, qual, (size_t)len); else for (int i = 0; i < len; ++i) d[i] = qual[len - 1 - i]; d += len; }
        else *d++ = '*';
        c.p = d;
    };
    if (h.type == 0) { c.lit("\t4\t*\t0\t0\t*\t*\t0\t0\t"); put_seq(h.strand); c.ch('\n'); used += (size_t)(c.p - c0); return; }
    int flag = 0;
    if (h.pos + span - ct->offset > ct->len) flag |= 4;      // bridges two reference sequences
    if (h.strand) flag |= 16;
    c.ch('\t'); c.num(flag); c.ch('\t'); c.mem(ct->name.data(), ct->name.size()); c.ch('\t');
    c.num((long)(h.pos - ct->offset + 1)); c.ch('\t'); c.num(h.mapq); c.ch('\t');
    c.cigar(h.n_cigar, h.cigar, len);
    c.lit("\t*\t0\t0\t");
    put_seq(h.strand);
    char XT = "NURM"[h.type];
    if (nn > 10) XT = 'N';
    c.lit("\tXT:A:"); c.ch(XT); c.lit("\tNM:i:"); c.num(nm);
    if (nn) { c.lit("\tXN:i:"); c.num(nn); }
    c.lit("\tX0:i:"); c.num(h.c1);
    if (h.c1 <= opt.max_top2) { c.lit("\tX1:i:"); c.num(h.c2); }
    c.lit("\tXM:i:"); c.num(h.n_mm); c.lit("\tXO:i:"); c.num(h.n_gapo); c.lit("\tXG:i:"); c.num(h.n_gapo + h.n_gape);
    c.lit("\tMD:Z:"); c.mem(md.data(), md.size());
    if (h.n_multi) c.lit("\tXA:Z:");
    used += (size_t)(c.p - c0);
    if (h.n_multi) xa_cur(b, h, len, o, used);
    *room(o, used, 1) = '\n'; ++used;
}

// ---- the same record as a BAM record (ps_map_to_bam: no SAM text in between).  Field for field what sam_line prints and
// ps_bam.cpp's encode_line would make of it: tests/test_bam.py compares the two routes record by record.
static inline void b_put32(std::string &o, uint32_t v) { char c[4] = {(char)(v & 0xff), (char)((v >> 8) & 0xff), (char)((v >> 16) & 0xff), (char)(v >> 24)}; o.append(c, 4); }
static inline void b_put16(std::string &o, uint32_t v) { char c[2] = {(char)(v & 0xff), (char)((v >> 8) & 0xff)}; o.append(c, 2); }
static inline void b_tag_int(std::string &o, const char *tag, long v)          // the smallest type that holds it, as htslib / encode_line
{
    o.push_back(tag[0]); o.push_back(tag[1]);
    if (v < 0) {
        if (v >= -128) { o.push_back('c'); o.push_back((char)(int8_t)v); }
        else if (v >= -32768) { o.push_back('s'); b_put16(o, (uint32_t)(uint16_t)(int16_t)v); }
        else { o.push_back('i'); b_put32(o, (uint32_t)(int32_t)v); }
    } else if (v <= 255) { o.push_back('C'); o.push_back((char)(uint8_t)v); }
    else if (v <= 65535) { o.push_back('S'); b_put16(o, (uint32_t)v); }
    else { o.push_back('I'); b_put32(o, (uint32_t)v); }
}
// false: below the MAPQ filter (not stored)
static bool bam_record(const Batch &b, int64_t g, int min_mapq, std::string &o, BamRec &r)
{
    const ReadSet &rs = b.rs; const RefSeq &ref = b.ctx->ix.ref; const Options &opt = b.ctx->opt;
    Hit h; b.hit_of(g, h);
    const int mapq = h.type == 0 ? 0 : h.mapq;
    if (mapq < min_mapq) return false;
    const int len = rs.len[g];
    const uint8_t *seq = rs.seq.data() + rs.off[g];
    const char *qual = rs.has_qual ? rs.qual.data() + rs.off[g] : nullptr;
    size_t nl; const char *nm_ = rs.name(g, nl);
    if (nl > 254) throw Error("read name longer than 254 characters");
    int seqid = -1, flag = 4, nn = 0, n_cig = 0; int64_t pos = -1, end = 0;
    uint32_t cig[PS_MAX_CIGAR + 1];
    if (h.type != 0) {
        const int span = (int)ref_span(h.n_cigar, h.cigar, len);
        nn = ref.cnt_ambi(h.pos, span, &seqid);
        const Contig &ct = ref.contigs[seqid];
        flag = (h.pos + span - ct.offset > ct.len ? 4 : 0) | (h.strand ? 16 : 0);
        pos = h.pos - ct.offset;
        if (h.n_cigar) { for (int j = 0; j < h.n_cigar; ++j) { const uint32_t op = h.cigar[j] & 0xfu; cig[j] = (h.cigar[j] & ~0xfu) | (op == 3 ? 4u : op); } n_cig = h.n_cigar; }
        else { cig[0] = (uint32_t)len << 4; n_cig = 1; }
        end = pos + (span > 0 ? span : 1);
    }
    const size_t start = o.size();
    b_put32(o, 0);
    b_put32(o, (uint32_t)seqid); b_put32(o, (uint32_t)(int32_t)pos);
    o.push_back((char)(uint8_t)(nl + 1)); o.push_back((char)(uint8_t)mapq);
    b_put16(o, (uint32_t)bam_reg2bin(pos, end));
    b_put16(o, (uint32_t)n_cig); b_put16(o, (uint32_t)flag);
    b_put32(o, (uint32_t)len); b_put32(o, 0xFFFFFFFFu); b_put32(o, 0xFFFFFFFFu); b_put32(o, 0);
    o.append(nm_, nl); o.push_back('\0');
    for (int j = 0; j < n_cig; ++j) b_put32(o, cig[j]);
    static const uint8_t NIB[5] = {1, 2, 4, 8, 15}, NIB_RC[5] = {8, 4, 2, 1, 15};
    const bool rc = h.strand != 0;
    {
        const size_t at = o.size(), nb = (size_t)(len + 1) / 2;
        o.resize(at + nb + (size_t)len);
        uint8_t *d = reinterpret_cast<uint8_t *>(&o[at]);
        for (int i = 0; i < len; i += 2) {
            const uint8_t hi = rc ? NIB_RC[seq[len - 1 - i]] : NIB[seq[i]];
            const uint8_t lo = i + 1 < len ? (rc ? NIB_RC[seq[len - 2 - i]] : NIB[seq[i + 1]]) : 0;
            d[i >> 1] = (uint8_t)(hi << 4 | lo);
        }
        d += nb;
        if (!qual) std::memset(d, 0xff, (size_t)len);
        else if (!rc) for (int i = 0; i < len; ++i) d[i] = (uint8_t)(qual[i] - 33);
        else for (int i = 0; i < len; ++i) d[i] = (uint8_t)(qual[len - 1 - i] - 33);
    }
    if (h.type != 0) {
        uint8_t tmp_small[256]; std::vector<uint8_t> tmp_big;
        uint8_t *tmp = tmp_small;
        if (len > 256) { tmp_big.resize((size_t)len); tmp = tmp_big.data(); }
        const uint8_t *oriented = seq;
        if (rc) { for (int i = 0; i < len; ++i) { uint8_t c = seq[len - 1 - i]; tmp[i] = c > 3 ? c : (uint8_t)(3 - c); } oriented = tmp; }
        static thread_local std::string md; md.clear(); int nm = 0;
        cal_md(ref, h.n_cigar, h.cigar, len, h.pos, oriented, md, nm);
        char XT = "NURM"[h.type];
        if (nn > 10) XT = 'N';
        o.append("XTA", 3); o.push_back(XT);
        b_tag_int(o, "NM", nm);
        if (nn) b_tag_int(o, "XN", nn);
        b_tag_int(o, "X0", h.c1);
        if (h.c1 <= opt.max_top2) b_tag_int(o, "X1", h.c2);
        b_tag_int(o, "XM", h.n_mm); b_tag_int(o, "XO", h.n_gapo); b_tag_int(o, "XG", h.n_gapo + h.n_gape);
        o.append("MDZ", 3); o.append(md); o.push_back('\0');
        if (h.n_multi) { o.append("XAZ", 3); xa_text(b, h, len, o); o.push_back('\0'); }
    }
    const uint32_t bs = (uint32_t)(o.size() - start - 4);
    o[start] = (char)(bs & 0xff); o[start + 1] = (char)((bs >> 8) & 0xff); o[start + 2] = (char)((bs >> 16) & 0xff); o[start + 3] = (char)(bs >> 24);
    r.ref = seqid; r.pos = (int32_t)pos; r.end = (int32_t)end; r.flag = (uint32_t)flag; r.off = start; r.len = o.size() - start; r.part = 0;
    return true;
}
// the records of a located batch that pass the MAPQ filter as BAM records: one buffer per host thread, the buffers in input order
void batch_bam_records(const Batch &b, int min_mapq, int threads, std::vector<std::string> &enc, std::vector<std::vector<BamRec>> &recs)
{
    if (!b.located) throw Error("BAM records before locate");
    const size_t N = (size_t)b.rs.n;
    const int nt = par_threads(N, threads);
    enc.assign((size_t)nt, std::string()); recs.assign((size_t)nt, std::vector<BamRec>());
    par_for(N, threads, [&](size_t g0, size_t g1, int t) {
        std::string &o = enc[t]; o.reserve((g1 - g0) * 176);
        const uint8_t *pac = b.ctx->ix.ref.pac_data();
        for (size_t g = g0; g < g1; ++g) {
            if (g + 8 < g1 && !(b.h_class[g + 8] & PS_CLS_HOST) && b.h_fin[g + 8].type) __builtin_prefetch(pac + ((size_t)b.h_fin[g + 8].pos >> 2));   // as batch_write_sam
            BamRec r; if (bam_record(b, (int64_t)g, min_mapq, o, r)) recs[t].push_back(r);
        }
    });
}

void batch_profile_records(const Batch &b, int min_mapq, int threads, ProfRecords &out)
{
    if (!b.located) throw Error("profile records before locate");
    const ReadSet &rs = b.rs; const RefSeq &ref = b.ctx->ix.ref;
    const size_t N = (size_t)rs.n;
    const int nt = par_threads(N, threads);
    // pass 1: which reads are in the filtered file, and how much they hold
    std::vector<uint8_t> keep(N, 0);
    std::vector<size_t> t_rec(nt + 1, 0), t_cig(nt + 1, 0), t_base(nt + 1, 0);
    par_for(N, threads, [&](size_t g0, size_t g1, int t) {
        size_t nr = 0, nc = 0, nb = 0;
        for (size_t g = g0; g < g1; ++g) {
            Hit h; b.hit_of((int64_t)g, h);
            if (h.type == 0 || h.mapq < min_mapq) continue;
            const int len = rs.len[g];
            int seqid = 0;
            const int span = (int)ref_span(h.n_cigar, h.cigar, len);
            ref.cnt_ambi(h.pos, span, &seqid);
            const Contig &ct = ref.contigs[seqid];
            if (h.pos + span - ct.offset > ct.len) continue;          // bridges two sequences: the SAM line carries flag 4 (sam_line)
            keep[g] = 1; ++nr; nc += h.n_cigar ? (size_t)h.n_cigar : 1; nb += (size_t)len + ((size_t)len & 1);
        }
        t_rec[t] = nr; t_cig[t] = nc; t_base[t] = nb;
    });
    size_t r0 = out.n(), c0 = out.cigar.size(), b0 = out.seq.size() * 2;     // every record starts on a whole byte
    std::vector<size_t> br(nt + 1), bc(nt + 1), bb(nt + 1);
    br[0] = r0; bc[0] = c0; bb[0] = b0;
    for (int t = 0; t < nt; ++t) { br[t + 1] = br[t] + t_rec[t]; bc[t + 1] = bc[t] + t_cig[t]; bb[t + 1] = bb[t] + t_base[t]; }
    out.gpos.resize(br[nt]); out.l_seq.resize(br[nt]); out.flag.resize(br[nt]); out.cig_off.resize(br[nt]); out.n_cig.resize(br[nt]); out.seq_off.resize(br[nt]);
    out.cigar.resize(bc[nt]); out.seq.resize(bb[nt] / 2);
    // pass 2: fill, every thread its own range
    static const uint8_t NIB[5] = {1, 2, 4, 8, 15}, NIB_RC[5] = {8, 4, 2, 1, 15};
    par_for(N, threads, [&](size_t g0, size_t g1, int t) {          // the same ranges as in pass 1
        size_t r = br[t], c = bc[t], bs = bb[t];
        for (size_t g = g0; g < g1; ++g) {
            if (!keep[g]) continue;
            Hit h; b.hit_of((int64_t)g, h);
            const int len = rs.len[g];
            const uint8_t *seq = rs.seq.data() + rs.off[g];
            out.gpos[r] = h.pos; out.l_seq[r] = len; out.flag[r] = h.strand ? 16u : 0u;
            out.cig_off[r] = (uint32_t)c; out.seq_off[r] = (uint64_t)bs;
            if (h.n_cigar) { for (int j = 0; j < h.n_cigar; ++j) { const uint32_t op = h.cigar[j] & 0xfu; out.cigar[c++] = (h.cigar[j] & ~0xfu) | (op == 3 ? 4u : op); } out.n_cig[r] = (uint32_t)h.n_cigar; }
            else { out.cigar[c++] = (uint32_t)len << 4; out.n_cig[r] = 1; }
            uint8_t *d = out.seq.data() + bs / 2;
            for (int i = 0; i < len; i += 2) {
                const uint8_t hi = h.strand ? NIB_RC[seq[len - 1 - i]] : NIB[seq[i]];
                const uint8_t lo = i + 1 < len ? (h.strand ? NIB_RC[seq[len - 2 - i]] : NIB[seq[i + 1]]) : 0;
                d[i >> 1] = (uint8_t)(hi << 4 | lo);
            }
            bs += (size_t)len + ((size_t)len & 1);
            ++r;
        }
    });
}

// @SQ per reference sequence in FASTA order, then our @PG: what upstream's samse prints before the first record -- also when
// there is no record at all (bwa_print_sam_SQ runs before the read loop)
std::string sam_header(const RefSeq &ref, const char *pg_line)
{
    std::string h;
    for (const Contig &c : ref.contigs) { h += "@SQ\tSN:"; h += c.name; h += "\tLN:"; put_int(h, c.len); h.push_back('\n'); }
    if (pg_line && pg_line[0]) { h += pg_line; h += "\n"; }
    return h;
}

void batch_write_sam(Batch &b, const char *path, bool header, const char *pg_line, int threads, bool append, SamScratch *scratch)
{
    if (!b.located) throw Error("write_sam before locate");
    const int fd = ::open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0644);
    if (fd < 0) throw Error(std::string("cannot write ") + path);
    struct Closer { int fd; bool done = false; ~Closer() { if (!done) ::close(fd); } } closer{fd};
    off_t at = append ? ::lseek(fd, 0, SEEK_END) : 0;
    if (at < 0) throw Error(std::string("cannot seek in ") + path);
    auto put = [&](const char *p, size_t n, off_t where) {           // the whole buffer at its place in the file
        while (n) { const ssize_t w = ::pwrite(fd, p, n, where); if (w <= 0) return false; p += w; n -= (size_t)w; where += w; }
        return true;
    };
    if (header) {
        const std::string h = sam_header(b.ctx->ix.ref, pg_line);
        if (!put(h.data(), h.size(), at)) throw Error(std::string("short write on ") + path);
        at += (off_t)h.size();
    }
    const int64_t N = b.rs.n;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    // rounds of threads x 64k reads: every thread formats its range; the text of a round is then written -- each buffer at its
    // own offset (pwrite), by a few I/O threads side by side -- while the next round is formatted.  (One writer thread managed
    // ~1 GB/s and was the slowest stage of ps_map at 2 GB of SAM per 10 M reads.)
    const int64_t chunk = 1 << 16;
    SamScratch own;
    std::vector<std::string> *bufs = scratch ? scratch->bufs : own.bufs;
    for (int k = 0; k < 2; ++k) if (bufs[k].size() < (size_t)threads) bufs[k].resize((size_t)threads);
    std::vector<off_t> where[2] = {std::vector<off_t>((size_t)threads, 0), std::vector<off_t>((size_t)threads, 0)};
    std::vector<size_t> lens[2] = {std::vector<size_t>((size_t)threads, 0), std::vector<size_t>((size_t)threads, 0)};
    std::thread io; std::atomic<bool> io_ok{true};
    const int n_io = std::max(1, std::min(8, threads));
    int which = 0;
    static const bool verbose = std::getenv("PS_VERBOSE") != nullptr && std::atoi(std::getenv("PS_VERBOSE")) >= 2;
    double t_fmt = 0, t_wait = 0; const auto tw0 = std::chrono::steady_clock::now();
    for (int64_t base = 0; base < N; base += chunk * threads, which ^= 1) {
        const auto tf0 = std::chrono::steady_clock::now();
        std::vector<std::string> &out = bufs[which];          // the I/O threads may still hold the other set
        std::vector<size_t> &used = lens[which];
        auto fmt = [&](int t) {
            int64_t g0 = base + chunk * t, g1 = std::min(N, g0 + chunk);
            std::string &o = out[t];                           // storage: its size is what it can hold, used[t] what it does hold
            size_t u = 0;
            if (g0 < g1) room(o, 0, (size_t)(g1 - g0) * 224);
            const uint8_t *pac = b.ctx->ix.ref.pac_data();
            for (int64_t g = g0; g < g1; ++g) {
                if (g + 8 < g1 && !(b.h_class[g + 8] & PS_CLS_HOST) && b.h_fin[g + 8].type) __builtin_prefetch(pac + ((size_t)b.h_fin[g + 8].pos >> 2));   // the reference bases of a read further on (MD tag): a cache miss each
                sam_line(b, g, o, u);
            }
            used[t] = u;
        };
        { std::vector<std::thread> th; for (int t = 1; t < threads; ++t) th.emplace_back(fmt, t); fmt(0); for (auto &x : th) x.join(); }
        const auto tf1 = std::chrono::steady_clock::now();
        if (io.joinable()) io.join();
        t_fmt += std::chrono::duration<double>(tf1 - tf0).count(); t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf1).count();
        if (!io_ok) break;
        std::vector<off_t> &wh = where[which];
        for (int t = 0; t < threads; ++t) { wh[t] = at; at += (off_t)used[t]; }
        io = std::thread([&out, &wh, &used, &put, &io_ok, n_io, threads]() {
            auto part = [&](int k) { for (int t = k; t < threads; t += n_io) if (used[t] && !put(out[t].data(), used[t], wh[t])) io_ok = false; };
            std::vector<std::thread> th; for (int k = 1; k < n_io; ++k) th.emplace_back(part, k); part(0); for (auto &x : th) x.join();
        });
    }
    const auto tl0 = std::chrono::steady_clock::now();
    if (io.joinable()) io.join();
    if (verbose) std::fprintf(stderr, "[parasuite-hip]     SAM text of %lld reads: %.0f ms (formatting on %d threads %.0f ms, waiting for the previous round's pwrite %.0f ms, last round's pwrite %.0f ms), %.0f MB\n", (long long)N,
                              1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count(), threads, 1e3 * t_fmt, 1e3 * t_wait, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count(), at / 1048576.0);
    if (!io_ok) throw Error(std::string("short write on ") + path);
    closer.done = true;
    if (::close(fd) != 0) throw Error(std::string("cannot close ") + path);
}

}  // namespace ps
