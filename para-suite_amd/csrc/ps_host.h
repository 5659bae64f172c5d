// ps_host.h -- internal host-side interface of libparasuite_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "ps_types.h"
#include "ps_model.h"
#include "ps_kernels.h"

namespace ps {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define PS_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    throw ps::Error(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// The product has no CPU path: every entry point that computes calls this first.
void require_device(int device);

template <class T> struct DevBuf {
    T *p = nullptr; size_t n = 0; bool owned = true;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), owned(o.owned) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; owned = o.owned; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; n = 0; owned = true; }
    void alloc(size_t count)                 // n is committed only when the allocation succeeded (a failed hipMalloc leaves an empty buffer)
    { release(); if (count) { T *q = nullptr; PS_HIP(hipMalloc((void **)&q, count * sizeof(T))); p = q; } n = count; }
    void adopt(T *ptr, size_t count) { release(); p = ptr; n = count; owned = false; }
    void zero(hipStream_t s = 0) { if (n) PS_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
    void upload(const T *src, size_t count, hipStream_t s = 0) { PS_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s)); }
    void download(T *dst, size_t count, hipStream_t s = 0) const { PS_HIP(hipMemcpyAsync(dst, p, count * sizeof(T), hipMemcpyDeviceToHost, s)); }
};

// ---- reference sequence metadata (what upstream keeps in .ann/.amb/.pac) ----
struct Contig { std::string name, anno; int64_t offset; int32_t len; int32_t n_ambs; };
struct Hole { int64_t offset; int32_t len; char amb; };
struct RefSeq {
    std::vector<Contig> contigs; std::vector<Hole> holes;
    std::vector<uint8_t> pac;   // forward strand, 2 bit (index build, ps_ctx_from_blobs)
    // index_load maps the .pac FILE instead (read-only, shared with the page cache): a copy of hg19's 775 MB took 0.1 s to make and
    // 0.1 s to give back at the end of every ps_map call
    std::shared_ptr<const void> pac_map; const uint8_t *pac_view = nullptr;
    const uint8_t *pac_data() const { return pac_view ? pac_view : pac.data(); }
    int64_t l_pac = 0;
    int pos2rid(int64_t pos_f) const;
    int cnt_ambi(int64_t pos_f, int len, int *ref_id) const;
};
void load_fasta(const char *path, RefSeq &ref);

// 48-bit LCG of POSIX drand48/lrand48 (the generator upstream bwa seeds with 11)
struct Rng48 {
    uint64_t x;
    explicit Rng48(long seed = 11) { x = (((uint64_t)(uint32_t)seed) << 16) | 0x330E; }
    uint64_t step() { x = (x * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL; return x; }
    double drand() { return (double)step() * (1.0 / 281474976710656.0); }
    long lrand() { return (long)(step() >> 17); }
    void jump(uint64_t t);      // advance by t draws in O(log t)
};

struct Index {
    RefSeq ref;
    DevBuf<OccBlock> blocks; DevBuf<uint32_t> sa; DevBuf<uint8_t> pac;   // sa: n_sa low words, then the bit-32 plane
    DevBuf<uint32_t> jump; int jump_levels = 0;                          // jump table (ps_core.h): built on the device after every build / load, never saved
    IndexView view;
    double build_ms = 0;
    int sa_rounds = 0;
    void refresh_view();
    static size_t sa_words(size_t n_sa) { return n_sa + (n_sa + 31) / 32; }
    size_t device_bytes() const { return blocks.n * sizeof(OccBlock) + sa.n * sizeof(uint32_t) + pac.n + jump.n * sizeof(uint32_t); }
};
void index_build(const char *fa, Index &ix, hipStream_t s);        // GPU suffix sorting (ps_index.hip)
void index_build_jump(Index &ix, hipStream_t s);                    // the view must be complete (blocks, primary, L2); PS_JUMP_LEVELS overrides the depth
void index_save(const Index &ix, const std::string &prefix);
void index_load(const std::string &prefix, Index &ix, hipStream_t s);
// a second copy of a resident index on another device, over xGMI (hipMemcpyPeerAsync); the caller has made dst_device current
void index_clone(const Index &src, int src_device, Index &dst, int dst_device, hipStream_t s);
bool index_files_exist(const std::string &prefix);
void index_load_pac(const std::string &prefix, Index &ix, hipStream_t s);   // .ann + .pac only (error-profile stage)
std::string index_meta_serialize(const Index &ix);
void index_meta_deserialize(const std::string &blob, Index &ix);   // fills ref + view scalars, no device data

// ---- error-profile estimation from a mapping (ps_profile.hip; what ErrorProfiling.inferErrorProfile computes) ----
struct ProfileCounts {                 // raw counts, read orientation; [pos][ref base][read base]
    int max_len = 0;
    std::vector<unsigned long long> conv, ins, del;     // conv: max_len*16; ins/del: per alignment column (forward strand), max_len
    unsigned long long n_records = 0, n_processed = 0, n_unmapped = 0, n_duplicate = 0, n_start_zero = 0, n_indel_reads = 0, n_skipped = 0;
};
// alignment records as the counting kernel takes them (BAM conventions: CIGAR words len<<4|op with MIDNSHP=X, bases as nibbles
// =ACMGRSVTWYHKDBN in the orientation of the SAM record); gpos = start on the packed forward strand, < 0: not counted
struct ProfRecords {
    std::vector<int64_t> gpos; std::vector<int32_t> l_seq; std::vector<uint32_t> flag, cig_off, n_cig, cigar;
    std::vector<uint64_t> seq_off; std::vector<uint8_t> seq;
    size_t n() const { return gpos.size(); }
};
// device-side totals that several batches of records add to (the fused first pass adds one batch per piece of the input)
struct Index;
class ProfileAccum {
public:
    ProfileAccum(int device, const Index &ix /* pac and holes; resident on `device` */, int max_len);
    ~ProfileAccum();
    void add(const ProfRecords &r);
    void finish(ProfileCounts &out);          // conv / ins / del and the kernel's counters; the caller fills the record statistics
private:
    struct Impl; Impl *p;
};
void error_profile_count(const char *mapping_sam_or_bam, const char *ref_prefix, int max_len, int device, int threads, ProfileCounts &out);
void error_profile_write(const ProfileCounts &c, const std::string &out_prefix);    // <out_prefix>.errorprofile / .indelprofile
std::string java_double_to_string(double v);                                        // java.lang.Double.toString

}  // namespace ps
