// ps_index.hip -- `bwa index` replacement: FASTA -> 2-bit pac, BWT of
// forward+reverse-complement text, 64-byte Occ blocks, SA sampled every 32 rows.
//
// Call site replaced: /root/reference/src/src/mapping/PARAsuiteMapping.java:45-55
// (index if <ref>.bwt is missing).  The suffix array is built ON THE GPU by
// prefix doubling over hipcub radix sorts (keys and ranks stay in HBM; a 1 Gbp
// genome needs ~70 GB of the 288 GB); there is no host suffix sorter.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
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
void load_fasta(const char *path, RefSeq &ref)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) throw Error(std::string("cannot open reference ") + path);
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)sz + 1);
    if (sz && std::fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); throw Error(std::string("short read on ") + path); }
    std::fclose(f);
    const size_t n = (size_t)sz;
    Rng48 rng(11);
    ref = RefSeq();
    ref.pac.assign(n / 4 + 2, 0);
    size_t i = 0;
    while (i < n) {
        while (i < n && buf[i] != '>') ++i;
        if (i >= n) break;
        size_t s = ++i;
        while (i < n && !std::isspace((unsigned char)buf[i])) ++i;
        Contig c;
        c.name.assign(buf.data() + s, i - s);
        size_t e = i; while (e < n && buf[e] != '\n') ++e;
        size_t cs = i; while (cs < e && std::isspace((unsigned char)buf[cs])) ++cs;
        size_t ce = e; while (ce > cs && std::isspace((unsigned char)buf[ce - 1])) --ce;
        c.anno = ce > cs ? std::string(buf.data() + cs, ce - cs) : std::string("(null)");
        i = e;
        c.offset = ref.contigs.empty() ? 0 : ref.contigs.back().offset + ref.contigs.back().len;
        c.n_ambs = 0;
        int lasts = 0; int32_t pos = 0; bool open_hole = false;
        for (; i < n && buf[i] != '>'; ++i) {
            int ch = (unsigned char)buf[i];
            if (!std::isgraph(ch)) continue;
            int code = nt4(ch);
            if (code >= 4) {
                if (open_hole && lasts == ch) ++ref.holes.back().len;
                else { ref.holes.push_back(Hole{c.offset + pos, 1, (char)ch}); ++c.n_ambs; open_hole = true; }
                code = (int)(rng.lrand() & 3);
            }
            lasts = ch;
            ref.pac[(size_t)ref.l_pac >> 2] |= (uint8_t)(code << ((~ref.l_pac & 3) << 1));
            ++ref.l_pac; ++pos;
        }
        c.len = pos;
        ref.contigs.push_back(c);
    }
    if (ref.contigs.empty()) throw Error(std::string("no sequences in ") + path);
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
typedef unsigned long long u64;

__global__ void k_expand_text(const uint8_t *pac, uint8_t *T, u64 l_pac)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < l_pac; i += (u64)gridDim.x * blockDim.x) {
        int b = pac_base(pac, (bwtint)i);
        T[i] = (uint8_t)b;
        T[2 * l_pac - 1 - i] = (uint8_t)(3 - b);     // reverse complement strand appended
    }
}
// first sort key: 27 base-5 digits (symbol+1, 0 beyond the text) -> suffixes that reach '$' order first
__global__ void k_init_keys(const uint8_t *T, u64 n, u64 *key, bwtint *sa)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (u64)gridDim.x * blockDim.x) {
        u64 k = 0;
#pragma unroll
        for (int j = 0; j < 27; ++j) k = k * 5 + (i + j < n ? (u64)T[i + j] + 1 : 0);
        key[i] = k; sa[i] = (bwtint)i;
    }
}
__global__ void k_group_flags(const u64 *key, u64 N, bwtint *gstart, unsigned long long *n_groups)
{
    unsigned long long local = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        bool head = i == 0 || key[i] != key[i - 1];
        gstart[i] = head ? (bwtint)i : 0;
        local += head;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(n_groups, local);
}
__global__ void k_scatter_rank(const bwtint *sa, const bwtint *gstart, u64 N, bwtint *rank)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) rank[sa[i]] = gstart[i];
}
__global__ void k_double_keys(const bwtint *sa, const bwtint *rank, u64 N, u64 h, u64 *key)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        u64 p = sa[i];
        key[i] = ((u64)rank[p] << 32) | (p + h < N ? (u64)rank[p + h] : 0ull);
    }
}
__global__ void k_find_primary(const bwtint *sa, u64 N, bwtint *primary)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x)
        if (sa[i] == 0) *primary = (bwtint)i;
}
// stored BWT symbol j (the '$' row skipped) = T[SA[row]-1]
__global__ void k_bwt_syms(const bwtint *sa, const uint8_t *T, u64 n, bwtint primary, uint8_t *B)
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
__global__ void k_sample_sa(const bwtint *sa, uint32_t n_sa, int intv, bwtint *out)
{
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n_sa; t += (u64)gridDim.x * blockDim.x)
        out[t] = t == 0 ? 0xFFFFFFFFu : sa[t * (u64)intv];
}

void Index::refresh_view()
{
    view.blocks = blocks.p; view.sa = sa.p; view.pac = pac.p;
    view.l_pac = (bwtint)ref.l_pac; view.seq_len = (bwtint)(2 * ref.l_pac);
    view.n_blocks = (uint32_t)blocks.n; view.n_sa = (uint32_t)sa.n; view.sa_intv = 32;
}

static const int GRID = 256 * 8, BLK = 256;

void index_build(const char *fa, Index &ix, hipStream_t s)
{
    auto t0 = std::chrono::steady_clock::now();
    load_fasta(fa, ix.ref);
    const u64 l_pac = (u64)ix.ref.l_pac, n = 2 * l_pac, N = n + 1;
    if (N >= 0xFFFFFFFFull) throw Error("reference too large for the 32-bit index of this build (2*l_pac must be < 2^32-1)");
    ix.pac.alloc(ix.ref.pac.size());
    ix.pac.upload(ix.ref.pac.data(), ix.ref.pac.size(), s);
    DevBuf<uint8_t> T; T.alloc(n + 32);
    hipLaunchKernelGGL(k_expand_text, dim3(GRID), dim3(BLK), 0, s, ix.pac.p, T.p, l_pac);
    DevBuf<u64> keyA, keyB; DevBuf<bwtint> saA, saB, rank, gstart; DevBuf<unsigned long long> cnt;
    keyA.alloc(N); keyB.alloc(N); saA.alloc(N); saB.alloc(N); rank.alloc(N); gstart.alloc(N); cnt.alloc(1);
    hipLaunchKernelGGL(k_init_keys, dim3(GRID), dim3(BLK), 0, s, T.p, n, keyA.p, saA.p);
    size_t tmp_sort = 0, tmp_scan = 0;
    PS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, keyA.p, keyB.p, saA.p, saB.p, (size_t)N, 0, 64, s));
    PS_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, tmp_scan, gstart.p, gstart.p, hipcub::Max(), (size_t)N, s));
    DevBuf<uint8_t> tmp; tmp.alloc(tmp_sort > tmp_scan ? tmp_sort : tmp_scan);
    size_t tb = tmp.n;
    PS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, keyA.p, keyB.p, saA.p, saB.p, (size_t)N, 0, 63, s));
    ix.sa_rounds = 0;
    for (u64 h = 27;; h *= 2) {
        // sorted keys in keyB, suffix order in saB
        cnt.zero(s);
        hipLaunchKernelGGL(k_group_flags, dim3(GRID), dim3(BLK), 0, s, keyB.p, N, gstart.p, cnt.p);
        unsigned long long groups = 0;
        cnt.download(&groups, 1, s);
        PS_HIP(hipStreamSynchronize(s));
        ++ix.sa_rounds;
        if (groups == N) break;
        if (h > N) throw Error("suffix sorting did not converge");
        tb = tmp.n;
        PS_HIP(hipcub::DeviceScan::InclusiveScan(tmp.p, tb, gstart.p, gstart.p, hipcub::Max(), (size_t)N, s));
        hipLaunchKernelGGL(k_scatter_rank, dim3(GRID), dim3(BLK), 0, s, saB.p, gstart.p, N, rank.p);
        hipLaunchKernelGGL(k_double_keys, dim3(GRID), dim3(BLK), 0, s, saB.p, rank.p, N, h, keyA.p);
        PS_HIP(hipMemcpyAsync(saA.p, saB.p, N * sizeof(bwtint), hipMemcpyDeviceToDevice, s));
        tb = tmp.n;
        PS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, keyA.p, keyB.p, saA.p, saB.p, (size_t)N, 0, 64, s));
    }
    keyA.release(); keyB.release(); rank.release(); gstart.release(); saA.release();
    // BWT, Occ blocks, sampled SA
    DevBuf<bwtint> dprim; dprim.alloc(1);
    hipLaunchKernelGGL(k_find_primary, dim3(GRID), dim3(BLK), 0, s, saB.p, N, dprim.p);
    bwtint primary = 0;
    dprim.download(&primary, 1, s);
    PS_HIP(hipStreamSynchronize(s));
    DevBuf<uint8_t> B; B.alloc(n + 1);
    hipLaunchKernelGGL(k_bwt_syms, dim3(GRID), dim3(BLK), 0, s, saB.p, T.p, n, primary, B.p);
    const uint32_t n_blocks = (uint32_t)(n / PS_BLK_SYMS + 1);
    DevBuf<uint32_t> c[4], cs[4];
    for (int j = 0; j < 4; ++j) { c[j].alloc(n_blocks); cs[j].alloc(n_blocks); }
    hipLaunchKernelGGL(k_block_counts, dim3(GRID), dim3(BLK), 0, s, B.p, n, n_blocks, c[0].p, c[1].p, c[2].p, c[3].p);
    size_t tmp2 = 0;
    PS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, c[0].p, cs[0].p, (size_t)n_blocks, s));
    if (tmp2 > tmp.n) tmp.alloc(tmp2);
    uint32_t last_c[4], last_s[4];
    for (int j = 0; j < 4; ++j) {
        tb = tmp.n;
        PS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, c[j].p, cs[j].p, (size_t)n_blocks, s));
        PS_HIP(hipMemcpyAsync(&last_c[j], c[j].p + (n_blocks - 1), 4, hipMemcpyDeviceToHost, s));
        PS_HIP(hipMemcpyAsync(&last_s[j], cs[j].p + (n_blocks - 1), 4, hipMemcpyDeviceToHost, s));
    }
    ix.blocks.alloc(n_blocks);
    hipLaunchKernelGGL(k_pack_blocks, dim3(GRID), dim3(BLK), 0, s, B.p, n, n_blocks, cs[0].p, cs[1].p, cs[2].p, cs[3].p, ix.blocks.p);
    const uint32_t n_sa = (uint32_t)((n + 32) / 32);
    ix.sa.alloc(n_sa);
    hipLaunchKernelGGL(k_sample_sa, dim3(GRID), dim3(BLK), 0, s, saB.p, n_sa, 32, ix.sa.p);
    PS_HIP(hipStreamSynchronize(s));
    ix.view.primary = primary;
    ix.view.L2[0] = 0;
    for (int j = 0; j < 4; ++j) ix.view.L2[j + 1] = ix.view.L2[j] + last_c[j] + last_s[j];
    ix.refresh_view();
    if (ix.view.L2[4] != (bwtint)n) throw Error("index build: symbol counts do not add up");
    ix.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// ------------------------------------------------------------- files ---------
// <ref>.bwt (header + Occ blocks), <ref>.sa, <ref>.pac, <ref>.ann (contigs + holes), this project's formats.
static const char MAGIC_BWT[8] = {'P', 'S', 'B', 'W', 'T', '0', '1', 0};

std::string index_meta_serialize(const Index &ix)
{
    std::ostringstream o;
    o << "PSANN01\n" << ix.ref.l_pac << ' ' << ix.ref.contigs.size() << ' ' << ix.ref.holes.size() << '\n';
    o << ix.view.seq_len << ' ' << ix.view.primary << ' ' << ix.view.L2[0] << ' ' << ix.view.L2[1] << ' ' << ix.view.L2[2] << ' '
      << ix.view.L2[3] << ' ' << ix.view.L2[4] << ' ' << ix.blocks.n << ' ' << ix.sa.n << ' ' << ix.pac.n << '\n';
    for (const Contig &c : ix.ref.contigs) o << c.name << '\t' << c.offset << '\t' << c.len << '\t' << c.n_ambs << '\t' << c.anno << '\n';
    for (const Hole &h : ix.ref.holes) o << h.offset << ' ' << h.len << ' ' << (int)(unsigned char)h.amb << '\n';
    return o.str();
}
void index_meta_deserialize(const std::string &blob, Index &ix)
{
    std::istringstream in(blob);
    std::string magic; std::getline(in, magic);
    if (magic != "PSANN01") throw Error("bad index metadata");
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
template <class T> static void read_dev(const std::string &path, const char *magic8, DevBuf<T> &d, hipStream_t s)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw Error("cannot open " + path);
    char mg[8]; uint64_t cnt = 0;
    if (std::fread(mg, 1, 8, f) != 8 || std::memcmp(mg, magic8, 8) != 0 || std::fread(&cnt, 8, 1, f) != 1) { std::fclose(f); throw Error("bad header in " + path); }
    std::vector<T> h(cnt);
    if (cnt && std::fread(h.data(), sizeof(T), cnt, f) != cnt) { std::fclose(f); throw Error("truncated " + path); }
    std::fclose(f);
    d.alloc(cnt);
    d.upload(h.data(), cnt, s);
    PS_HIP(hipStreamSynchronize(s));
}

void index_save(const Index &ix, const std::string &prefix)
{
    write_dev(prefix + ".bwt", MAGIC_BWT, ix.blocks);
    write_dev(prefix + ".sa", "PSSA0001", ix.sa);
    write_dev(prefix + ".pac", "PSPAC001", ix.pac);
    std::ofstream a(prefix + ".ann", std::ios::binary);
    a << index_meta_serialize(ix);
    if (!a.good()) throw Error("cannot write " + prefix + ".ann");
}

void index_load(const std::string &prefix, Index &ix, hipStream_t s)
{
    std::ifstream a(prefix + ".ann", std::ios::binary);
    if (!a.good()) throw Error("cannot open " + prefix + ".ann (run ps_index first)");
    std::stringstream ss; ss << a.rdbuf();
    index_meta_deserialize(ss.str(), ix);
    read_dev(prefix + ".bwt", MAGIC_BWT, ix.blocks, s);
    read_dev(prefix + ".sa", "PSSA0001", ix.sa, s);
    read_dev(prefix + ".pac", "PSPAC001", ix.pac, s);
    ix.ref.pac.resize(ix.pac.n);
    PS_HIP(hipMemcpy(ix.ref.pac.data(), ix.pac.p, ix.pac.n, hipMemcpyDeviceToHost));
    bwtint primary = ix.view.primary; bwtint L2[5]; std::memcpy(L2, ix.view.L2, sizeof L2);
    ix.refresh_view();
    ix.view.primary = primary; std::memcpy(ix.view.L2, L2, sizeof L2);
}

}  // namespace ps
