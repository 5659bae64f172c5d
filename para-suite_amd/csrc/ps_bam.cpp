// ps_bam.cpp -- SAM text -> BAM (BGZF), MAPQ filter, coordinate / name sort and .bai index, on SAM or BAM input.
//
// §8f rank 3: what /root/reference/src/src/mapping/PARAsuiteMapping.java:102-133 (`samtools view -bS`,
// `samtools view -q <mapq> -b`) and Mapping.java:85-108 (`samtools sort`, `samtools index`) spawn four
// processes and three rewrites of the data for.  Formats follow the public SAM/BAM specification (SAMv1:
// BGZF blocks of at most 64 KiB with the BC extra field, BAM records, binning index with 16 kb linear index);
// integer tags take the smallest type as htslib's SAM parser does.  Host code only (zlib), no device work.
#include <zlib.h>
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include "ps_bam.h"

namespace ps {
namespace {

struct Err : std::runtime_error { explicit Err(const std::string &m) : std::runtime_error(m) {} };

inline void put32(std::string &o, uint32_t v) { char b[4] = {(char)(v & 0xff), (char)((v >> 8) & 0xff), (char)((v >> 16) & 0xff), (char)(v >> 24)}; o.append(b, 4); }
inline void put16(std::string &o, uint16_t v) { char b[2] = {(char)(v & 0xff), (char)(v >> 8)}; o.append(b, 2); }
inline void put64(std::string &o, uint64_t v) { put32(o, (uint32_t)v); put32(o, (uint32_t)(v >> 32)); }

// UCSC binning scheme (SAMv1 §5.3)
inline int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

typedef BamRec Rec;        // encoded record: bytes [off, off+len) of its part

struct Field { const char *p; size_t n; };

static int base4(char c)
{
    switch (c) {
        case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3;
        case 'G': case 'g': return 4; case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7;
        case 'T': case 't': return 8; case 'W': case 'w': return 9; case 'Y': case 'y': return 10; case 'H': case 'h': return 11;
        case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14; default: return 15;
    }
}
static long to_long(Field f) { return std::strtol(std::string(f.p, f.n).c_str(), nullptr, 10); }

// one SAM alignment line -> BAM record appended to o; false if filtered out
static bool encode_line(const char *line, size_t n, const std::map<std::string, int> &ref_id, int min_mapq, std::string &o, Rec &r)
{
    Field f[11]; size_t i = 0; int k = 0;
    while (k < 11) {
        const char *e = (const char *)std::memchr(line + i, '\t', n - i);
        size_t j = e ? (size_t)(e - line) : n;
        f[k].p = line + i; f[k].n = j - i; ++k;
        i = j < n ? j + 1 : n;
        if (!e && k < 11) throw Err("SAM line with fewer than 11 fields: " + std::string(line, std::min<size_t>(n, 80)));
    }
    const long flag = to_long(f[1]), pos1 = to_long(f[3]), mapq = to_long(f[4]), pnext1 = to_long(f[7]), tlen = to_long(f[8]);
    if (mapq < min_mapq) return false;
    int ref = -1, rnext = -1;
    if (!(f[2].n == 1 && f[2].p[0] == '*')) {
        auto it = ref_id.find(std::string(f[2].p, f[2].n));
        if (it == ref_id.end()) throw Err("reference name not in the header: " + std::string(f[2].p, f[2].n));
        ref = it->second;
    }
    if (f[6].n == 1 && f[6].p[0] == '=') rnext = ref;
    else if (!(f[6].n == 1 && f[6].p[0] == '*')) {
        auto it = ref_id.find(std::string(f[6].p, f[6].n));
        if (it == ref_id.end()) throw Err("mate reference name not in the header");
        rnext = it->second;
    }
    // CIGAR
    std::vector<uint32_t> cig;
    int64_t ref_len = 0;
    if (!(f[5].n == 1 && f[5].p[0] == '*')) {
        uint32_t num = 0; bool any = false;
        for (size_t c = 0; c < f[5].n; ++c) {
            const char ch = f[5].p[c];
            if (ch >= '0' && ch <= '9') { num = num * 10 + (uint32_t)(ch - '0'); any = true; continue; }
            static const char ops[] = "MIDNSHP=X";
            const char *w = std::strchr(ops, ch);
            if (!w || !any) throw Err("bad CIGAR: " + std::string(f[5].p, f[5].n));
            const int op = (int)(w - ops);
            cig.push_back((num << 4) | (uint32_t)op);
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += num;
            num = 0; any = false;
        }
    }
    const int32_t pos = (int32_t)pos1 - 1;
    const int64_t end = pos + (ref_len > 0 ? ref_len : 1);
    const bool has_seq = !(f[9].n == 1 && f[9].p[0] == '*');
    const uint32_t l_seq = has_seq ? (uint32_t)f[9].n : 0;
    const size_t start = o.size();
    put32(o, 0);                                            // block_size, patched below
    put32(o, (uint32_t)ref); put32(o, (uint32_t)pos);
    o.push_back((char)(uint8_t)(f[0].n + 1)); o.push_back((char)(uint8_t)mapq);
    put16(o, (uint16_t)reg2bin(pos, end));
    put16(o, (uint16_t)cig.size()); put16(o, (uint16_t)flag);
    put32(o, l_seq); put32(o, (uint32_t)rnext); put32(o, (uint32_t)((int32_t)pnext1 - 1)); put32(o, (uint32_t)(int32_t)tlen);
    if (f[0].n > 254) throw Err("read name longer than 254 characters");
    o.append(f[0].p, f[0].n); o.push_back('\0');
    for (uint32_t c : cig) put32(o, c);
    for (uint32_t c = 0; c < l_seq; c += 2) {
        const int hi = base4(f[9].p[c]), lo = c + 1 < l_seq ? base4(f[9].p[c + 1]) : 0;
        o.push_back((char)(uint8_t)((hi << 4) | lo));
    }
    if (f[10].n == 1 && f[10].p[0] == '*') o.append((size_t)l_seq, (char)0xff);
    else {
        if (f[10].n != l_seq) throw Err("SEQ and QUAL of different length");
        for (uint32_t c = 0; c < l_seq; ++c) o.push_back((char)(uint8_t)(f[10].p[c] - 33));
    }
    // optional fields TAG:TYPE:VALUE
    while (i < n) {
        const char *e = (const char *)std::memchr(line + i, '\t', n - i);
        const size_t j = e ? (size_t)(e - line) : n;
        const char *t = line + i; const size_t tn = j - i;
        i = j < n ? j + 1 : n;
        if (tn == 0) continue;
        if (tn < 5 || t[2] != ':' || t[4] != ':') throw Err("bad optional field: " + std::string(t, tn));
        o.push_back(t[0]); o.push_back(t[1]);
        const char ty = t[3]; const char *v = t + 5; const size_t vn = tn - 5;
        if (ty == 'A') { o.push_back('A'); o.push_back(vn ? v[0] : ' '); }
        else if (ty == 'i') {
            const long long x = std::strtoll(std::string(v, vn).c_str(), nullptr, 10);
            if (x < 0) {
                if (x >= -128) { o.push_back('c'); o.push_back((char)(int8_t)x); }
                else if (x >= -32768) { o.push_back('s'); put16(o, (uint16_t)(int16_t)x); }
                else { o.push_back('i'); put32(o, (uint32_t)(int32_t)x); }
            } else {
                if (x <= 255) { o.push_back('C'); o.push_back((char)(uint8_t)x); }
                else if (x <= 65535) { o.push_back('S'); put16(o, (uint16_t)x); }
                else { o.push_back('I'); put32(o, (uint32_t)x); }
            }
        } else if (ty == 'f') { o.push_back('f'); float fl = std::strtof(std::string(v, vn).c_str(), nullptr); uint32_t u; std::memcpy(&u, &fl, 4); put32(o, u); }
        else if (ty == 'Z' || ty == 'H') { o.push_back(ty); o.append(v, vn); o.push_back('\0'); }
        else throw Err(std::string("optional field type not supported: ") + ty);
    }
    const uint32_t bs = (uint32_t)(o.size() - start - 4);
    o[start] = (char)(bs & 0xff); o[start + 1] = (char)((bs >> 8) & 0xff); o[start + 2] = (char)((bs >> 16) & 0xff); o[start + 3] = (char)(bs >> 24);
    r.ref = ref; r.pos = pos; r.end = (int32_t)end; r.flag = (uint32_t)flag; r.off = start; r.len = o.size() - start;
    return true;
}

// one BGZF block (gzip member with the BC extra field) of at most 0xff00 input bytes
static void bgzf_block(const char *src, size_t n, int level, std::string &out)
{
    static const unsigned char head[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    std::vector<unsigned char> buf(n + n / 100 + 64);
    z_stream zs; std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw Err("deflateInit2 failed");
    zs.next_in = (Bytef *)const_cast<char *>(src); zs.avail_in = (uInt)n;
    zs.next_out = buf.data(); zs.avail_out = (uInt)buf.size();
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); throw Err("deflate failed"); }
    const size_t cn = zs.total_out;
    deflateEnd(&zs);
    const size_t total = 18 + cn + 8;
    if (total > 65536) throw Err("BGZF block too large");
    out.append((const char *)head, 16);
    put16(out, (uint16_t)(total - 1));
    out.append((const char *)buf.data(), cn);
    put32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)src, (uInt)n));
    put32(out, (uint32_t)n);
}

template <class F> static void par(int n, int threads, F f)
{
    std::vector<std::thread> th; std::vector<std::string> err((size_t)n);
    int next = 0;
    while (next < n) {
        th.clear();
        for (int t = 0; t < threads && next < n; ++t, ++next) { const int id = next; th.emplace_back([&, id]() { try { f(id); } catch (const std::exception &e) { err[id] = e.what(); if (err[id].empty()) err[id] = "error"; } }); }
        for (auto &x : th) x.join();
    }
    for (auto &e : err) if (!e.empty()) throw Err(e);
}

}  // namespace

namespace {

struct Data {                       // a BAM in memory: header, reference table, encoded records in parts
    std::string text; std::vector<std::pair<std::string, uint32_t>> refs;
    std::vector<std::string> enc; std::vector<Rec> recs; uint64_t n_in = 0;
};

static void header_sorted(std::string &text, const char *so)
{
    if (text.compare(0, 3, "@HD") == 0) {
        const size_t e = text.find('\n');
        std::string hd = text.substr(0, e);
        const size_t at = hd.find("\tSO:");
        if (at != std::string::npos) { size_t q = hd.find('\t', at + 1); hd.erase(at, (q == std::string::npos ? hd.size() : q) - at); }
        hd += std::string("\tSO:") + so;
        text = hd + text.substr(e);
    } else text = std::string("@HD\tVN:1.6\tSO:") + so + "\n" + text;
}

// ---- SAM text -> Data
static void load_sam(const char *sam_path, int min_mapq, int threads, Data &d)
{
    FILE *f = std::fopen(sam_path, "rb");
    if (!f) throw Err(std::string("cannot open ") + sam_path);
    std::fseek(f, 0, SEEK_END); const long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)sz + 1);
    if (sz && std::fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); throw Err(std::string("short read on ") + sam_path); }
    std::fclose(f);
    const size_t n = (size_t)sz; const char *b = buf.data();
    std::map<std::string, int> ref_id;
    size_t body = 0;
    while (body < n && b[body] == '@') {
        const char *e = (const char *)std::memchr(b + body, '\n', n - body);
        const size_t j = e ? (size_t)(e - b) : n;
        std::string line(b + body, j - body);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.compare(0, 3, "@SQ") == 0) {
            std::string name; long ln = 0; size_t p = 0;
            while (p < line.size()) {
                size_t q = line.find('\t', p); if (q == std::string::npos) q = line.size();
                if (line.compare(p, 3, "SN:") == 0) name = line.substr(p + 3, q - p - 3);
                else if (line.compare(p, 3, "LN:") == 0) ln = std::strtol(line.substr(p + 3, q - p - 3).c_str(), nullptr, 10);
                p = q + 1;
            }
            if (name.empty() || ln <= 0) throw Err("bad @SQ line: " + line);
            ref_id[name] = (int)d.refs.size(); d.refs.emplace_back(name, (uint32_t)ln);
        }
        d.text += line; d.text.push_back('\n');
        body = j < n ? j + 1 : n;
    }
    // records, encoded in parallel over ranges of whole lines
    std::vector<size_t> cut(1, body);
    for (int t = 1; t < threads; ++t) {
        size_t at = body + (n - body) / (size_t)threads * (size_t)t;
        const char *e = at < n ? (const char *)std::memchr(b + at, '\n', n - at) : nullptr;
        at = e ? (size_t)(e - b) + 1 : n;
        if (at > cut.back() && at < n) cut.push_back(at);
    }
    cut.push_back(n);
    const int parts = (int)cut.size() - 1;
    d.enc.assign((size_t)parts, std::string()); std::vector<std::vector<Rec>> recs((size_t)parts);
    std::vector<uint64_t> n_in((size_t)parts, 0);
    par(parts, threads, [&](int t) {
        std::string &o = d.enc[t]; o.reserve((cut[t + 1] - cut[t]));
        size_t i = cut[t];
        while (i < cut[t + 1]) {
            const char *e = (const char *)std::memchr(b + i, '\n', cut[t + 1] - i);
            size_t j = e ? (size_t)(e - b) : cut[t + 1], j2 = j;
            if (j2 > i && b[j2 - 1] == '\r') --j2;
            if (j2 > i) {
                Rec r; r.part = t;
                ++n_in[t];
                if (encode_line(b + i, j2 - i, ref_id, min_mapq, o, r)) recs[t].push_back(r);
            }
            i = j + 1;
        }
    });
    size_t m = 0; for (auto &v : recs) m += v.size();
    d.recs.reserve(m);
    for (int t = 0; t < parts; ++t) { d.recs.insert(d.recs.end(), recs[t].begin(), recs[t].end()); d.n_in += n_in[t]; }
}

// ---- BGZF file -> uncompressed bytes; block starts (compressed offset, uncompressed offset)
struct Blocks { std::string data; std::vector<std::pair<uint64_t, uint64_t>> starts; uint64_t file_bytes = 0; };
static void inflate_bgzf(const char *path, int threads, Blocks &out)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) throw Err(std::string("cannot open ") + path);
    std::fseek(f, 0, SEEK_END); const long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> raw((size_t)sz);
    if (sz && std::fread(raw.data(), 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); throw Err(std::string("short read on ") + path); }
    std::fclose(f);
    out.file_bytes = (uint64_t)sz;
    struct B { size_t at, bsize; uint32_t isize; uint64_t u; };
    std::vector<B> bl; size_t at = 0; uint64_t u = 0;
    while (at < raw.size()) {
        if (at + 28 > raw.size() || raw[at] != 31 || raw[at + 1] != 139 || raw[at + 12] != 'B' || raw[at + 13] != 'C') throw Err(std::string("not a BGZF file: ") + path);
        const size_t bsize = (size_t)(raw[at + 16] | (raw[at + 17] << 8)) + 1;
        if (at + bsize > raw.size()) throw Err(std::string("truncated BGZF block in ") + path);
        const unsigned char *t = raw.data() + at + bsize - 4;
        const uint32_t isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        bl.push_back(B{at, bsize, isize, u});
        u += isize; at += bsize;
    }
    out.data.assign((size_t)u, '\0');
    out.starts.clear();
    for (const B &b : bl) out.starts.emplace_back((uint64_t)b.at, b.u);
    const int T = threads;
    par(T, T, [&](int t) {
        for (size_t k = (size_t)t; k < bl.size(); k += (size_t)T) {
            const B &b = bl[k];
            if (!b.isize) continue;
            z_stream zs; std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) throw Err("inflateInit2 failed");
            zs.next_in = raw.data() + b.at + 18; zs.avail_in = (uInt)(b.bsize - 26);
            zs.next_out = (Bytef *)&out.data[(size_t)b.u]; zs.avail_out = b.isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.total_out != b.isize) throw Err(std::string("corrupt BGZF block in ") + path);
        }
    });
}
static uint32_t rd32(const std::string &s, size_t at) { const unsigned char *p = (const unsigned char *)s.data() + at; return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// ---- BAM file -> Data (records stay as they are; their place in the file is kept for indexing)
static void load_bam(const char *path, int threads, Data &d, Blocks *keep_blocks, std::vector<uint64_t> *rec_u)
{
    Blocks local; Blocks &bk = keep_blocks ? *keep_blocks : local;
    inflate_bgzf(path, threads, bk);
    const std::string &s = bk.data;
    if (s.size() < 12 || s.compare(0, 4, "BAM\1") != 0) throw Err(std::string("not a BAM file: ") + path);
    const uint32_t l_text = rd32(s, 4);
    d.text = s.substr(8, l_text);
    while (!d.text.empty() && d.text.back() == '\0') d.text.pop_back();
    size_t at = 8 + (size_t)l_text;
    const uint32_t n_ref = rd32(s, at); at += 4;
    for (uint32_t r = 0; r < n_ref; ++r) {
        const uint32_t ln = rd32(s, at);
        d.refs.emplace_back(s.substr(at + 4, ln ? ln - 1 : 0), rd32(s, at + 4 + ln));
        at += 8 + (size_t)ln;
    }
    d.enc.assign(1, std::string());
    d.enc[0].assign(s, at, std::string::npos);
    const std::string &e = d.enc[0];
    const size_t base_u = at;
    size_t p = 0;
    while (p + 4 <= e.size()) {
        const uint32_t bs = rd32(e, p);
        if (bs < 32 || p + 4 + bs > e.size()) throw Err(std::string("corrupt BAM record in ") + path);
        Rec r; r.part = 0; r.off = p; r.len = 4 + (size_t)bs;
        r.ref = (int32_t)rd32(e, p + 4); r.pos = (int32_t)rd32(e, p + 8);
        const uint32_t w = rd32(e, p + 12), fl = rd32(e, p + 16);
        const uint32_t l_name = w & 0xff, n_cig = fl & 0xffff;
        r.flag = fl >> 16;
        int64_t ref_len = 0;
        const size_t cig_at = p + 36 + l_name;
        for (uint32_t c = 0; c < n_cig; ++c) { const uint32_t v = rd32(e, cig_at + 4 * c); const int op = (int)(v & 15); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += v >> 4; }
        r.end = (int32_t)(r.pos + (ref_len > 0 ? ref_len : 1));
        d.recs.push_back(r);
        if (rec_u) rec_u->push_back((uint64_t)(base_u + p));
        p += 4 + bs;
    }
    d.n_in = d.recs.size();
    if (!keep_blocks) { bk.data.clear(); bk.data.shrink_to_fit(); }
}
static int rec_mapq(const Data &d, const Rec &r) { return (int)((rd32(d.enc[r.part], r.off + 12) >> 8) & 0xff); }
static const char *rec_name(const Data &d, const Rec &r) { return d.enc[r.part].data() + r.off + 36; }

// read names as `samtools sort -n` orders them: digit runs compare as numbers
static int strnum_cmp(const char *a, const char *b)
{
    const unsigned char *pa = (const unsigned char *)a, *pb = (const unsigned char *)b;
    while (*pa && *pb) {
        if (std::isdigit(*pa) && std::isdigit(*pb)) {
            while (*pa == '0') ++pa;
            while (*pb == '0') ++pb;
            const unsigned char *ea = pa, *eb = pb;
            while (std::isdigit(*ea)) ++ea;
            while (std::isdigit(*eb)) ++eb;
            if (ea - pa != eb - pb) return (ea - pa) < (eb - pb) ? -1 : 1;
            for (; pa < ea; ++pa, ++pb) if (*pa != *pb) return *pa < *pb ? -1 : 1;
        } else {
            if (*pa != *pb) return *pa < *pb ? -1 : 1;
            ++pa; ++pb;
        }
    }
    return *pa ? 1 : (*pb ? -1 : 0);
}

// virtual offsets of the records of an existing layout, then the .bai
struct Layout { std::vector<uint64_t> vbeg, vend; };
static void write_bai(const std::string &bai_path, const Data &d, const Layout &lay)
{
    struct RefIdx { std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::vector<uint64_t> lin; uint64_t beg = 0, end = 0, n_map = 0, n_unmap = 0; bool any = false; };
    std::vector<RefIdx> idx(d.refs.size());
    uint64_t n_no_coor = 0;
    for (size_t i = 0; i < d.recs.size(); ++i) {
        const Rec &r = d.recs[i];
        if (r.ref < 0) { ++n_no_coor; continue; }
        if ((size_t)r.ref >= idx.size()) throw Err("record refers to a reference that is not in the header");
        RefIdx &x = idx[(size_t)r.ref];
        const uint64_t vb = lay.vbeg[i], ve = lay.vend[i];
        if (!x.any) { x.beg = vb; x.any = true; }
        x.end = ve;
        if (r.flag & 4) ++x.n_unmap; else ++x.n_map;
        const uint32_t bin = (uint32_t)reg2bin(r.pos, r.end);
        auto &ch = x.bins[bin];
        if (!ch.empty() && (ch.back().second >> 16 == vb >> 16 || ch.back().second == vb)) ch.back().second = ve;     // same compressed block or adjacent: extend
        else ch.emplace_back(vb, ve);
        const size_t w0 = (size_t)(r.pos >> 14), w1 = (size_t)((r.end - 1) >> 14);
        if (x.lin.size() <= w1) x.lin.resize(w1 + 1, 0);
        for (size_t w = w0; w <= w1; ++w) if (x.lin[w] == 0) x.lin[w] = vb;
    }
    std::string bai;
    bai.append("BAI\1", 4); put32(bai, (uint32_t)d.refs.size());
    for (RefIdx &x : idx) {
        put32(bai, (uint32_t)(x.bins.size() + (x.any ? 1 : 0)));
        for (auto &kv : x.bins) {
            put32(bai, kv.first); put32(bai, (uint32_t)kv.second.size());
            for (auto &c : kv.second) { put64(bai, c.first); put64(bai, c.second); }
        }
        if (x.any) { put32(bai, 37450u); put32(bai, 2u); put64(bai, x.beg); put64(bai, x.end); put64(bai, x.n_map); put64(bai, x.n_unmap); }
        for (size_t w = 1; w < x.lin.size(); ++w) if (x.lin[w] == 0) x.lin[w] = x.lin[w - 1];
        put32(bai, (uint32_t)x.lin.size());
        for (uint64_t v : x.lin) put64(bai, v);
    }
    put64(bai, n_no_coor);
    FILE *bi = std::fopen(bai_path.c_str(), "wb");
    if (!bi) throw Err("cannot write " + bai_path);
    bool ok = std::fwrite(bai.data(), 1, bai.size(), bi) == bai.size();
    ok = (std::fclose(bi) == 0) && ok;
    if (!ok) throw Err("short write on " + bai_path);
}

// ---- Data (records in d.recs order) -> BGZF file (+ .bai)
static void write_bam(Data &d, const char *bam_path, bool write_index, int threads, BamStats *stats, int level = 6)
{
    std::string head;
    head.append("BAM\1", 4); put32(head, (uint32_t)d.text.size()); head += d.text; put32(head, (uint32_t)d.refs.size());
    for (auto &r : d.refs) { put32(head, (uint32_t)r.first.size() + 1); head += r.first; head.push_back('\0'); put32(head, r.second); }
    const size_t BLK = 0xff00;
    const size_t head_blocks = (head.size() + BLK - 1) / BLK;           // own blocks: samtools flushes after the header too
    const std::vector<Rec> &all = d.recs;
    std::vector<uint64_t> uoff(all.size() + 1, 0);
    for (size_t i = 0; i < all.size(); ++i) uoff[i + 1] = uoff[i] + all[i].len;
    const uint64_t body_bytes = uoff[all.size()];
    const size_t body_blocks = (size_t)((body_bytes + BLK - 1) / BLK);
    std::string stream((size_t)body_bytes, '\0');
    {
        const int T = threads; const size_t per = (all.size() + (size_t)T - 1) / (size_t)T;
        par(T, T, [&](int t) { const size_t a0 = (size_t)t * per, a1 = std::min(all.size(), a0 + per); for (size_t i = a0; i < a1; ++i) std::memcpy(&stream[(size_t)uoff[i]], d.enc[all[i].part].data() + all[i].off, all[i].len); });
    }
    for (auto &e : d.enc) { e.clear(); e.shrink_to_fit(); }
    const size_t n_blocks = head_blocks + body_blocks;
    std::vector<std::string> comp(n_blocks);
    {
        const int T = threads;
        par(T, T, [&](int t) {
            for (size_t k = (size_t)t; k < n_blocks; k += (size_t)T) {
                if (k < head_blocks) { const size_t a0 = k * BLK; bgzf_block(head.data() + a0, std::min(BLK, head.size() - a0), level, comp[k]); }
                else { const size_t a0 = (k - head_blocks) * BLK; bgzf_block(stream.data() + a0, (size_t)std::min<uint64_t>(BLK, body_bytes - a0), level, comp[k]); }
            }
        });
    }
    std::vector<uint64_t> coff(n_blocks + 1, 0);
    for (size_t k = 0; k < n_blocks; ++k) coff[k + 1] = coff[k] + comp[k].size();
    static const unsigned char eof_block[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    FILE *o = std::fopen(bam_path, "wb");
    if (!o) throw Err(std::string("cannot write ") + bam_path);
    bool ok = true;
    for (size_t k = 0; k < n_blocks && ok; ++k) ok = std::fwrite(comp[k].data(), 1, comp[k].size(), o) == comp[k].size();
    ok = ok && std::fwrite(eof_block, 1, 28, o) == 28;
    ok = (std::fclose(o) == 0) && ok;
    if (!ok) throw Err(std::string("short write on ") + bam_path);
    if (stats) { stats->n_in = d.n_in; stats->n_out = all.size(); stats->bam_bytes = coff[n_blocks] + 28; }
    if (!write_index) return;
    auto voff = [&](uint64_t u) -> uint64_t {           // virtual file offset of body byte u
        if (u == body_bytes && u % BLK == 0) return coff[n_blocks] << 16;      // end of the last full block = start of the EOF block
        return (coff[head_blocks + (size_t)(u / BLK)] << 16) | (u % BLK);
    };
    Layout lay; lay.vbeg.resize(all.size()); lay.vend.resize(all.size());
    for (size_t i = 0; i < all.size(); ++i) { lay.vbeg[i] = voff(uoff[i]); lay.vend[i] = voff(uoff[i + 1]); }
    write_bai(std::string(bam_path) + ".bai", d, lay);
}

static void sort_records(Data &d, bool by_name)
{
    if (by_name)
        std::stable_sort(d.recs.begin(), d.recs.end(), [&](const Rec &x, const Rec &y) {
            const int c = strnum_cmp(rec_name(d, x), rec_name(d, y));
            if (c) return c < 0;
            return ((x.flag >> 6) & 3) < ((y.flag >> 6) & 3);             // first of pair before second
        });
    else
        std::stable_sort(d.recs.begin(), d.recs.end(), [](const Rec &x, const Rec &y) {
            const uint32_t a = (uint32_t)x.ref, c = (uint32_t)y.ref;      // -1 (no reference) sorts last
            return a != c ? a < c : x.pos < y.pos;
        });
}

}  // namespace

static int clamp_threads(int t) { return t < 1 ? 1 : (t > 64 ? 64 : t); }

int bam_reg2bin(int64_t beg, int64_t end) { return reg2bin(beg, end); }

// ---- records straight from memory (ps_map_to_bam)
struct BamSink::Impl {
    Data d; std::string path; bool sort = false, index = false; int threads = 1, level = 6;
    FILE *f = nullptr; uint64_t bytes = 0, n_out = 0; bool failed = false;
};
BamSink::BamSink(const std::string &header_text, const std::vector<std::pair<std::string, uint32_t>> &refs, const char *bam_path,
                 bool sort_by_coordinate, bool write_index, int threads, int level) : p(new Impl())
{
    if (write_index && !sort_by_coordinate) { delete p; p = nullptr; throw Err("a .bai index needs coordinate-sorted output"); }
    p->d.text = header_text; p->d.refs = refs; p->path = bam_path; p->sort = sort_by_coordinate; p->index = write_index;
    p->threads = clamp_threads(threads); p->level = level < 0 ? 0 : (level > 9 ? 9 : level);
    if (p->sort) { header_sorted(p->d.text, "coordinate"); return; }
    p->f = std::fopen(bam_path, "wb");
    if (!p->f) { const std::string m = std::string("cannot write ") + bam_path; delete p; p = nullptr; throw Err(m); }
    std::string head;
    head.append("BAM\1", 4); put32(head, (uint32_t)p->d.text.size()); head += p->d.text; put32(head, (uint32_t)p->d.refs.size());
    for (auto &r : p->d.refs) { put32(head, (uint32_t)r.first.size() + 1); head += r.first; head.push_back('\0'); put32(head, r.second); }
    const size_t BLK = 0xff00;
    for (size_t a0 = 0; a0 < head.size(); a0 += BLK) {
        std::string blk; bgzf_block(head.data() + a0, std::min(BLK, head.size() - a0), p->level, blk);
        if (std::fwrite(blk.data(), 1, blk.size(), p->f) != blk.size()) p->failed = true;
        p->bytes += blk.size();
    }
}
BamSink::~BamSink() { if (p) { if (p->f) std::fclose(p->f); delete p; } }
void BamSink::add(std::vector<std::string> &records, std::vector<std::vector<BamRec>> &recs, uint64_t n_in)
{
    p->d.n_in += n_in;
    if (p->sort) {
        for (size_t k = 0; k < records.size(); ++k) {
            const int part = (int)p->d.enc.size();
            for (BamRec &r : recs[k]) r.part = part;
            p->d.enc.push_back(std::move(records[k]));
            p->d.recs.insert(p->d.recs.end(), recs[k].begin(), recs[k].end());
        }
        return;
    }
    // every buffer is cut into its own BGZF blocks (a record may span two blocks, a block never spans two buffers); all blocks of
    // the call are compressed side by side
    const size_t BLK = 0xff00;
    std::vector<std::pair<size_t, size_t>> blk;              // (buffer, offset)
    for (size_t k = 0; k < records.size(); ++k) { p->n_out += recs[k].size(); for (size_t a0 = 0; a0 < records[k].size(); a0 += BLK) blk.emplace_back(k, a0); }
    std::vector<std::string> comp(blk.size());
    const int T = p->threads;
    par(T, T, [&](int t) { for (size_t j = (size_t)t; j < blk.size(); j += (size_t)T) { const std::string &src = records[blk[j].first]; bgzf_block(src.data() + blk[j].second, std::min(BLK, src.size() - blk[j].second), p->level, comp[j]); } });
    for (size_t j = 0; j < comp.size(); ++j) { if (std::fwrite(comp[j].data(), 1, comp[j].size(), p->f) != comp[j].size()) p->failed = true; p->bytes += comp[j].size(); }
    records.clear(); recs.clear();
}
void BamSink::finish(BamStats *stats)
{
    if (p->sort) {
        sort_records(p->d, false);
        write_bam(p->d, p->path.c_str(), p->index, p->threads, stats, p->level);
        return;
    }
    static const unsigned char eof_block[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = !p->failed && std::fwrite(eof_block, 1, 28, p->f) == 28;
    ok = (std::fclose(p->f) == 0) && ok;
    p->f = nullptr;
    if (!ok) throw Err(std::string("short write on ") + p->path);
    if (stats) { stats->n_in = p->d.n_in; stats->n_out = p->n_out; stats->bam_bytes = p->bytes + 28; }
}

void sam_to_bam(const char *sam_path, const char *bam_path, int min_mapq, bool sort_by_coordinate, bool write_index, int threads, BamStats *stats)
{
    threads = clamp_threads(threads);
    if (write_index && !sort_by_coordinate) throw Err("a .bai index needs coordinate-sorted output");
    Data d;
    load_sam(sam_path, min_mapq, threads, d);
    if (sort_by_coordinate) { header_sorted(d.text, "coordinate"); sort_records(d, false); }
    write_bam(d, bam_path, write_index, threads, stats);
}

// `samtools view -q <mapq> -b in.bam -o out.bam`
void bam_view(const char *in_bam, const char *out_bam, int min_mapq, int threads, BamStats *stats)
{
    threads = clamp_threads(threads);
    Data d;
    load_bam(in_bam, threads, d, nullptr, nullptr);
    if (min_mapq > 0) {
        std::vector<Rec> keep; keep.reserve(d.recs.size());
        for (const Rec &r : d.recs) if (rec_mapq(d, r) >= min_mapq) keep.push_back(r);
        d.recs.swap(keep);
    }
    write_bam(d, out_bam, false, threads, stats);
}

// `samtools sort [-n] in.bam -o out.bam`
void bam_sort(const char *in_bam, const char *out_bam, bool by_name, int threads, BamStats *stats)
{
    threads = clamp_threads(threads);
    Data d;
    load_bam(in_bam, threads, d, nullptr, nullptr);
    header_sorted(d.text, by_name ? "queryname" : "coordinate");
    sort_records(d, by_name);
    write_bam(d, out_bam, false, threads, stats);
}

// `samtools index in.bam`: <in.bam>.bai for a coordinate-sorted file, the file itself is not rewritten
void bam_index(const char *bam, int threads)
{
    threads = clamp_threads(threads);
    Data d; Blocks bk; std::vector<uint64_t> rec_u;
    load_bam(bam, threads, d, &bk, &rec_u);
    for (size_t i = 1; i < d.recs.size(); ++i) {
        const uint32_t a = (uint32_t)d.recs[i - 1].ref, c = (uint32_t)d.recs[i].ref;
        if (a > c || (a == c && d.recs[i - 1].pos > d.recs[i].pos)) throw Err(std::string("not sorted by coordinate: ") + bam);
    }
    auto voff = [&](uint64_t u) -> uint64_t {           // uncompressed offset -> (compressed block start << 16) | offset inside
        size_t lo = 0, hi = bk.starts.size();
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (bk.starts[mid].second <= u) lo = mid; else hi = mid; }
        // an offset that is exactly a block boundary belongs to the start of the later block (skip empty blocks except the last)
        return (bk.starts[lo].first << 16) | (u - bk.starts[lo].second);
    };
    const uint64_t total = bk.data.size();
    auto voff_end = [&](uint64_t u) -> uint64_t {       // the end of the data is "the last data block, past its last byte"
        if (u < total || u == 0) return voff(u);
        size_t lo = 0, hi = bk.starts.size();
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (bk.starts[mid].second <= u - 1) lo = mid; else hi = mid; }
        return (bk.starts[lo].first << 16) | (u - bk.starts[lo].second);
    };
    Layout lay; lay.vbeg.resize(d.recs.size()); lay.vend.resize(d.recs.size());
    for (size_t i = 0; i < d.recs.size(); ++i) { lay.vbeg[i] = voff(rec_u[i]); lay.vend[i] = voff_end(rec_u[i] + d.recs[i].len); }
    write_bai(std::string(bam) + ".bai", d, lay);
}

void load_alignments(const char *path, int threads, AlnTable &out)
{
    threads = clamp_threads(threads);
    unsigned char mg[2] = {0, 0};
    { FILE *f = std::fopen(path, "rb"); if (!f) throw Err(std::string("cannot open ") + path); const size_t g = std::fread(mg, 1, 2, f); (void)g; std::fclose(f); }
    Data d;
    if (mg[0] == 31 && mg[1] == 139) load_bam(path, threads, d, nullptr, nullptr); else load_sam(path, -1, threads, d);
    out = AlnTable();
    out.refs = d.refs;
    const size_t n = d.recs.size();
    out.ref.resize(n); out.pos.resize(n); out.l_seq.resize(n); out.flag.resize(n); out.cig_off.resize(n); out.n_cig.resize(n); out.seq_off.resize(n);
    uint64_t bases = 0; uint32_t ops = 0;
    for (size_t i = 0; i < n; ++i) {
        const Rec &r = d.recs[i]; const std::string &e = d.enc[r.part];
        const uint32_t w = rd32(e, r.off + 12), fl = rd32(e, r.off + 16), l_seq = rd32(e, r.off + 20);
        out.ref[i] = r.ref; out.pos[i] = r.pos; out.flag[i] = fl >> 16; out.l_seq[i] = (int32_t)l_seq;
        out.n_cig[i] = fl & 0xffff; out.cig_off[i] = ops; out.seq_off[i] = bases;
        ops += fl & 0xffff; bases += (l_seq + 1) / 2 * 2;          // records start on a byte
        (void)w;
    }
    out.cigar.resize(ops); out.seq.assign((size_t)(bases / 2), 0);
    par(threads, threads, [&](int t) {
        for (size_t i = (size_t)t; i < n; i += (size_t)threads) {
            const Rec &r = d.recs[i]; const std::string &e = d.enc[r.part];
            const uint32_t l_name = rd32(e, r.off + 12) & 0xff;
            const size_t cig_at = r.off + 36 + l_name, seq_at = cig_at + 4 * (size_t)out.n_cig[i];
            for (uint32_t c = 0; c < out.n_cig[i]; ++c) out.cigar[out.cig_off[i] + c] = rd32(e, cig_at + 4 * c);
            std::memcpy(&out.seq[(size_t)(out.seq_off[i] / 2)], e.data() + seq_at, (size_t)((out.l_seq[i] + 1) / 2));
        }
    });
}

}  // namespace ps
