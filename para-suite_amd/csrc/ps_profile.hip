// ps_profile.hip -- error-profile estimation from the first mapping pass.
//
// Replaces utils.errorprofile.ErrorProfiling.inferErrorProfile (/root/reference/src/src/utils/errorprofile/
// ErrorProfiling.java:100-631; counting loop :145-409, output :504-531 and :545-591), the single-threaded stage between
// the two mapping passes of a `--refine` run (Main.java:320-340).  Same counts, same two files:
//   <mapping>.errorprofile  four lines, row = reference base A C G T, column = read base, P(read | ref) pooled over the
//                           positions, in READ orientation, every value Double.toString + TAB (NaN for a base never seen)
//   <mapping>.indelprofile  "<ins>\t<del>", no newline: the mean over the alignment columns with a non-zero rate of
//                           (gaps starting at the column) / (bases counted at that read position)
// The records are parsed on the host (SAM text or BAM: ps_bam.cpp), the counting is one kernel over the records (one
// record per lane, block-level histograms in LDS, 64-bit totals in HBM), the reference comes from the index's packed
// forward strand + its hole table (a hole = any non-ACGT letter of the FASTA: never counted, as in the Java where
// calculateArrayPos returns -1 for it, :634-664).  Quirks of the Java that are kept because they shape the numbers:
// positions are columns of the alignment as rebuilt from the CIGAR only when read and reference span differ in length
// (:196-290; equal-length spans are compared base by base whatever the CIGAR says), an insertion's own bases and a
// deletion's reference bases are never counted, gap counts are booked at column (columns so far + q), q = 1..length,
// in forward-strand coordinates while the base counts they are divided by are in read orientation (:247-272, :553-570).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include "ps_host.h"
#include "ps_bam.h"

namespace ps {

struct ProfArgs {
    const int32_t *ref_off_lo; const int32_t *ref_off_hi;   // per record: global start on the packed forward strand (64-bit as two words), < 0 hi: skip
    const int32_t *l_seq; const uint32_t *flag; const uint32_t *cig_off; const uint32_t *n_cig; const uint32_t *cigar;
    const uint64_t *seq_off; const uint8_t *seq;
    const uint8_t *pac; const int64_t *hole_off; const int32_t *hole_len; int n_holes;
    int n_records, max_len;
    unsigned long long *conv, *ins, *del, *stat;              // stat: processed, indel reads, skipped, too long
};

__device__ __forceinline__ int prof_read_code(const uint8_t *seq, uint64_t base)     // BAM nibble -> 0..3, -1 otherwise
{
    const int nib = (seq[base >> 1] >> ((~base & 1u) << 2)) & 15;
    return nib == 1 ? 0 : (nib == 2 ? 1 : (nib == 4 ? 2 : (nib == 8 ? 3 : -1)));
}
struct ProfRef {              // reference bases of one record with its holes
    const uint8_t *pac; const int64_t *hole_off; const int32_t *hole_len; int n_holes, h;
    __device__ void seek(int64_t p)           // first hole that ends behind p
    {
        int lo = 0, hi = n_holes;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (hole_off[mid] + hole_len[mid] <= p) lo = mid + 1; else hi = mid; }
        h = lo;
    }
    __device__ int at(int64_t p) const
    {
        int k = h;
        while (k < n_holes && hole_off[k] + hole_len[k] <= p) ++k;
        if (k < n_holes && hole_off[k] <= p) return -1;
        return (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
    }
};

// one record per lane.  col = column of the rebuilt alignment (forward strand); a counted pair goes to position
// strand ? width-1-col : col with both bases complemented on the reverse strand (ErrorProfiling.java:301-306).
__global__ void __launch_bounds__(256) k_profile(ProfArgs a)
{
    extern __shared__ unsigned int sm[];
    unsigned int *s_conv = sm, *s_ins = sm + a.max_len * 16, *s_del = s_ins + a.max_len;
    for (int i = threadIdx.x; i < a.max_len * 18; i += blockDim.x) sm[i] = 0;
    __syncthreads();
    unsigned long long n_proc = 0, n_indel = 0, n_skip = 0, n_long = 0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < a.n_records; r += gridDim.x * blockDim.x) {
        if (a.ref_off_hi[r] < 0) continue;                                  // unmapped / duplicate / start 0: counted by the host
        const int64_t g0 = ((int64_t)a.ref_off_hi[r] << 32) | (uint32_t)a.ref_off_lo[r];
        const int L = a.l_seq[r];
        const uint32_t *cg = a.cigar + a.cig_off[r]; const int nc = (int)a.n_cig[r];
        int R = 0;
        for (int c = 0; c < nc; ++c) { const int op = (int)(cg[c] & 15u), len = (int)(cg[c] >> 4); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) R += len; }
        if (R < 1) R = 1;                                                    // htsjdk: alignment end = start for an empty span
        ++n_proc;
        const int width = L > R ? L : R;
        if (width > a.max_len) { ++n_long; continue; }                      // the Java would fail here (array bound): reported as an error by the host
        const bool strand = (a.flag[r] & 16u) != 0;
        const uint64_t sb = a.seq_off[r];
        ProfRef rf{a.pac, a.hole_off, a.hole_len, a.n_holes, 0};
        rf.seek(g0);
        if (L == R) {                                                       // spans of equal length: base by base, the CIGAR is not looked at
            for (int c = 0; c < L; ++c) {
                const int pr = rf.at(g0 + c), pd = prof_read_code(a.seq, sb + (uint64_t)c);
                if (pr >= 0 && pd >= 0) atomicAdd(&s_conv[(strand ? L - 1 - c : c) * 16 + (strand ? (3 - pr) * 4 + (3 - pd) : pr * 4 + pd)], 1u);
            }
            continue;
        }
        ++n_indel;
        // pass 1: gap columns are booked whatever follows; a match block that leaves the arrays marks the read as skipped
        bool skip = false;
        { int pm = 0, pref = 0, prd = 0;
          for (int c = 0; c < nc; ++c) {
              const int op = (int)(cg[c] & 15u), len = (int)(cg[c] >> 4);
              if (op == 0 || op == 7 || op == 8) { if (pm + len > width || pref + len > R || prd + len > L) skip = true; pm += len; pref += len; prd += len; }
              else if (op == 3) { pref += len; prd += len; }
              else if (op == 1) { pm += len; prd += len; for (int q = 1; q <= len; ++q) if (pm + q < a.max_len) atomicAdd(&s_ins[pm + q], 1u); }
              else if (op == 2) { pm += len; pref += len; for (int q = 1; q <= len; ++q) if (pm + q < a.max_len) atomicAdd(&s_del[pm + q], 1u); }
          } }
        if (skip) { ++n_skip; continue; }
        // pass 2: the match columns
        { int pm = 0, pref = 0, prd = 0;
          for (int c = 0; c < nc; ++c) {
              const int op = (int)(cg[c] & 15u), len = (int)(cg[c] >> 4);
              if (op == 0 || op == 7 || op == 8) {
                  for (int z = 0; z < len; ++z) {
                      const int pr = rf.at(g0 + pref + z), pd = prof_read_code(a.seq, sb + (uint64_t)(prd + z)), col = pm + z;
                      if (pr >= 0 && pd >= 0) atomicAdd(&s_conv[(strand ? width - 1 - col : col) * 16 + (strand ? (3 - pr) * 4 + (3 - pd) : pr * 4 + pd)], 1u);
                  }
                  pm += len; pref += len; prd += len;
              } else if (op == 3) { pref += len; prd += len; }
              else if (op == 1) { pm += len; prd += len; }
              else if (op == 2) { pm += len; pref += len; }
          } }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.max_len * 16; i += blockDim.x) if (s_conv[i]) atomicAdd(&a.conv[i], (unsigned long long)s_conv[i]);
    for (int i = threadIdx.x; i < a.max_len; i += blockDim.x) { if (s_ins[i]) atomicAdd(&a.ins[i], (unsigned long long)s_ins[i]); if (s_del[i]) atomicAdd(&a.del[i], (unsigned long long)s_del[i]); }
    if (n_proc) atomicAdd(&a.stat[0], n_proc);
    if (n_indel) atomicAdd(&a.stat[1], n_indel);
    if (n_skip) atomicAdd(&a.stat[2], n_skip);
    if (n_long) atomicAdd(&a.stat[3], n_long);
}

struct ProfileAccum::Impl {
    int device, max_len; hipStream_t s = nullptr; const uint8_t *pac;
    DevBuf<int64_t> d_hoff; DevBuf<int32_t> d_hlen; int n_holes = 0;
    DevBuf<unsigned long long> d_acc; size_t n_acc = 0;
};
ProfileAccum::ProfileAccum(int device, const Index &ix, int max_len) : p(new Impl())
{
    if (max_len < 1 || max_len > 4096) { delete p; throw Error("error profile: maximum read length out of range"); }
    p->device = device; p->max_len = max_len; p->pac = ix.pac.p;
    try {
        require_device(device);
        PS_HIP(hipStreamCreateWithFlags(&p->s, hipStreamNonBlocking));
        std::vector<int64_t> hoff(ix.ref.holes.size()); std::vector<int32_t> hlen(ix.ref.holes.size());
        for (size_t h = 0; h < ix.ref.holes.size(); ++h) { hoff[h] = ix.ref.holes[h].offset; hlen[h] = ix.ref.holes[h].len; }
        p->n_holes = (int)hoff.size();
        p->d_hoff.alloc(std::max<size_t>(1, hoff.size())); p->d_hlen.alloc(std::max<size_t>(1, hlen.size()));
        if (!hoff.empty()) { p->d_hoff.upload(hoff.data(), hoff.size(), p->s); p->d_hlen.upload(hlen.data(), hlen.size(), p->s); }
        p->n_acc = (size_t)max_len * 18 + 4;
        p->d_acc.alloc(p->n_acc); p->d_acc.zero(p->s);
        PS_HIP(hipStreamSynchronize(p->s));
    } catch (...) { if (p->s) (void)hipStreamDestroy(p->s); delete p; throw; }
}
ProfileAccum::~ProfileAccum() { if (p->s) (void)hipStreamDestroy(p->s); delete p; }
void ProfileAccum::add(const ProfRecords &t)
{
    const size_t n = t.n();
    if (!n) return;
    if (n > 0x7fffffffull) throw Error("error profile: more than 2^31 records in one call");
    require_device(p->device);
    hipStream_t s = p->s;
    std::vector<int32_t> lo(n), hi(n);
    for (size_t i = 0; i < n; ++i) { const int64_t g = t.gpos[i]; lo[i] = g < 0 ? 0 : (int32_t)(uint32_t)(g & 0xffffffffll); hi[i] = g < 0 ? -1 : (int32_t)(g >> 32); }
    DevBuf<int32_t> d_lo, d_hi, d_lseq; DevBuf<uint32_t> d_flag, d_coff, d_nc, d_cig; DevBuf<uint64_t> d_soff; DevBuf<uint8_t> d_seq;
    auto up = [&](auto &d, const auto &v) { d.alloc(std::max<size_t>(1, v.size())); if (!v.empty()) d.upload(v.data(), v.size(), s); };
    up(d_lo, lo); up(d_hi, hi); up(d_lseq, t.l_seq); up(d_flag, t.flag); up(d_coff, t.cig_off); up(d_nc, t.n_cig); up(d_cig, t.cigar); up(d_soff, t.seq_off); up(d_seq, t.seq);
    const int max_len = p->max_len;
    ProfArgs a;
    a.ref_off_lo = d_lo.p; a.ref_off_hi = d_hi.p; a.l_seq = d_lseq.p; a.flag = d_flag.p; a.cig_off = d_coff.p; a.n_cig = d_nc.p; a.cigar = d_cig.p;
    a.seq_off = d_soff.p; a.seq = d_seq.p; a.pac = p->pac; a.hole_off = p->d_hoff.p; a.hole_len = p->d_hlen.p; a.n_holes = p->n_holes;
    a.n_records = (int)n; a.max_len = max_len;
    a.conv = p->d_acc.p; a.ins = p->d_acc.p + (size_t)max_len * 16; a.del = a.ins + max_len; a.stat = a.del + max_len;
    const size_t lds = (size_t)max_len * 18 * sizeof(unsigned int);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_profile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = (int)std::min<size_t>(2048, (n + 255) / 256); if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_profile, dim3(blocks), dim3(256), lds, s, a);
    PS_HIP(hipGetLastError());
    PS_HIP(hipStreamSynchronize(s));          // the record arrays above are released on return
}
void ProfileAccum::finish(ProfileCounts &out)
{
    require_device(p->device);
    const int max_len = p->max_len;
    std::vector<unsigned long long> acc(p->n_acc);
    p->d_acc.download(acc.data(), p->n_acc, p->s);
    PS_HIP(hipStreamSynchronize(p->s));
    out.max_len = max_len;
    out.conv.assign(acc.begin(), acc.begin() + (size_t)max_len * 16);
    out.ins.assign(acc.begin() + (size_t)max_len * 16, acc.begin() + (size_t)max_len * 17);
    out.del.assign(acc.begin() + (size_t)max_len * 17, acc.begin() + (size_t)max_len * 18);
    out.n_processed = acc[(size_t)max_len * 18]; out.n_indel_reads = acc[(size_t)max_len * 18 + 1]; out.n_skipped = acc[(size_t)max_len * 18 + 2];
    if (acc[(size_t)max_len * 18 + 3]) throw Error("error profile: a read (or its reference span) is longer than the maximum read length given (the reference's arrays would overflow)");
}

void error_profile_count(const char *mapping, const char *ref_prefix, int max_len, int device, int threads, ProfileCounts &out)
{
    if (max_len < 1 || max_len > 4096) throw Error("error profile: maximum read length out of range");
    require_device(device);
    AlnTable t;
    try { load_alignments(mapping, threads, t); } catch (const std::exception &e) { throw Error(e.what()); }
    hipStream_t s; PS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct SG { hipStream_t s; ~SG() { (void)hipStreamDestroy(s); } } sg{s};
    Index ix;
    index_load_pac(ref_prefix, ix, s);
    std::map<std::string, int> cid;
    for (size_t c = 0; c < ix.ref.contigs.size(); ++c) cid[ix.ref.contigs[c].name] = (int)c;
    std::vector<int> ref_to_contig(t.refs.size(), -1);
    for (size_t r = 0; r < t.refs.size(); ++r) { auto it = cid.find(t.refs[r].first); if (it != cid.end()) ref_to_contig[r] = it->second; }
    const size_t n = t.n();
    out = ProfileCounts(); out.max_len = max_len;
    unsigned long long n_unmapped = 0, n_duplicate = 0, n_start_zero = 0;
    ProfRecords r;
    r.gpos.assign(n, -1);
    for (size_t i = 0; i < n; ++i) {
        if (t.flag[i] & 4u) { ++n_unmapped; continue; }                       // ErrorProfiling.java:155-158
        if (t.flag[i] & 1024u) { ++n_duplicate; continue; }                   // :159-162
        if (t.pos[i] < 0) { ++n_start_zero; continue; }                       // :163-166 (alignment start 0 = no position)
        if (t.ref[i] < 0 || (size_t)t.ref[i] >= ref_to_contig.size() || ref_to_contig[t.ref[i]] < 0) throw Error("error profile: a record names a sequence the reference does not have");
        r.gpos[i] = ix.ref.contigs[ref_to_contig[t.ref[i]]].offset + (int64_t)t.pos[i];
    }
    r.l_seq = std::move(t.l_seq); r.flag = std::move(t.flag); r.cig_off = std::move(t.cig_off); r.n_cig = std::move(t.n_cig); r.cigar = std::move(t.cigar);
    r.seq_off = std::move(t.seq_off); r.seq = std::move(t.seq);
    ProfileAccum acc(device, ix, max_len);
    acc.add(r);
    acc.finish(out);
    out.n_records = n; out.n_unmapped = n_unmapped; out.n_duplicate = n_duplicate; out.n_start_zero = n_start_zero;
}

// java.lang.Double.toString: the shortest decimal that reads back as the same double (the JDK 19+ definition; older JDKs
// print a longer digit string for a few values), plain notation with at least one fraction digit for 1e-3 <= |v| < 1e7,
// otherwise d.dddE<exp>
std::string java_double_to_string(double v)
{
    if (v != v) return "NaN";
    if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64]; int prec = 1;
    for (; prec <= 17; ++prec) { std::snprintf(buf, sizeof buf, "%.*e", prec - 1, v); if (std::strtod(buf, nullptr) == v) break; }
    std::string m(buf); const size_t ep = m.find('e');
    const int e10 = std::atoi(m.c_str() + ep + 1);
    std::string digits; bool neg = false;
    for (size_t i = 0; i < ep; ++i) { if (m[i] == '-') neg = true; else if (m[i] >= '0' && m[i] <= '9') digits.push_back(m[i]); }
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string o = neg ? "-" : "";
    const double av = std::fabs(v);
    if (av >= 1e-3 && av < 1e7) {
        if (e10 >= 0) {
            std::string ip = digits.substr(0, std::min(digits.size(), (size_t)e10 + 1));
            while ((int)ip.size() < e10 + 1) ip.push_back('0');
            std::string fp = digits.size() > (size_t)e10 + 1 ? digits.substr((size_t)e10 + 1) : "0";
            o += ip + "." + fp;
        } else o += "0." + std::string((size_t)(-e10 - 1), '0') + digits;
    } else {
        o += digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : "0") + "E" + std::to_string(e10);
    }
    return o;
}

// the two files of ErrorProfiling.java:504-531 and :545-591
void error_profile_write(const ProfileCounts &c, const std::string &out_prefix)
{
    const int ML = c.max_len;
    double tot[4][4] = {{0}}, base[4] = {0, 0, 0, 0};
    std::vector<double> per_pos((size_t)ML, 0.0);
    for (int i = 0; i < ML; ++i)
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 4; ++k) { const double x = (double)c.conv[(size_t)i * 16 + j * 4 + k]; tot[j][k] += x; base[j] += x; per_pos[i] += x; }
    {
        FILE *f = std::fopen((out_prefix + ".errorprofile").c_str(), "wb");
        if (!f) throw Error("cannot write " + out_prefix + ".errorprofile");
        for (int j = 0; j < 4; ++j) {
            for (int k = 0; k < 4; ++k) std::fprintf(f, "%s\t", java_double_to_string(tot[j][k] / base[j]).c_str());
            std::fputc('\n', f);
        }
        std::fclose(f);
    }
    double ins_all = 0, del_all = 0; int ins_zero = 0, del_zero = 0;
    for (int i = 0; i < ML; ++i) {
        if (per_pos[i] == 0.0) { ++ins_zero; ++del_zero; continue; }
        const double x = (double)c.ins[i] / per_pos[i], y = (double)c.del[i] / per_pos[i];
        if (x > 0) ins_all += x; else ++ins_zero;
        if (y > 0) del_all += y; else ++del_zero;
    }
    if (ML == ins_zero && ML == del_zero) ins_all = del_all = 0.0;
    else { ins_all = ins_all / (ML - ins_zero); del_all = del_all / (ML - del_zero); }
    FILE *f = std::fopen((out_prefix + ".indelprofile").c_str(), "wb");
    if (!f) throw Error("cannot write " + out_prefix + ".indelprofile");
    std::fprintf(f, "%s\t%s", java_double_to_string(ins_all).c_str(), java_double_to_string(del_all).c_str());
    std::fclose(f);
}

}  // namespace ps
