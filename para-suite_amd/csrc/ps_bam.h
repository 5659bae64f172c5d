// ps_bam.h -- SAM text -> BAM / sorted BAM + .bai, and the same operations on BAM input (host only, zlib).
#pragma once
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>
namespace ps {
struct BamStats { uint64_t n_in = 0, n_out = 0, bam_bytes = 0; };
// one encoded BAM record inside a buffer of records: reference id, 0-based position and end, flag, byte range [off, off + len)
struct BamRec { int32_t ref; int32_t pos; int32_t end; uint32_t flag; size_t off, len; int part; };
int bam_reg2bin(int64_t beg, int64_t end);                  // UCSC binning scheme (SAMv1 5.3)
// BAM records that were never SAM text (ps_map_to_bam: straight from the alignment records in memory): parts arrive in input
// order.  Unsorted output: a part is cut into BGZF blocks, compressed on `threads` threads and appended to the file at once, so the
// compression of one piece of the input runs while the next is searched.  Coordinate-sorted output (+ .bai): the parts are kept and
// sorted, compressed and written by finish().
class BamSink {
public:
    BamSink(const std::string &header_text, const std::vector<std::pair<std::string, uint32_t>> &refs, const char *bam_path,
            bool sort_by_coordinate, bool write_index, int threads, int level);
    ~BamSink();
    // buffers of records in input order (one per encoding thread); recs[k][i].off/len index into records[k]; both are consumed
    void add(std::vector<std::string> &records, std::vector<std::vector<BamRec>> &recs, uint64_t n_in);
    void finish(BamStats *stats);
private:
    struct Impl; Impl *p;
};
// all throw std::runtime_error; min_mapq: records with MAPQ below it are dropped (samtools view -q)
void sam_to_bam(const char *sam_path, const char *bam_path, int min_mapq, bool sort_by_coordinate, bool write_index, int threads, BamStats *stats);
void bam_view(const char *in_bam, const char *out_bam, int min_mapq, int threads, BamStats *stats);     // samtools view -q Q -b
void bam_sort(const char *in_bam, const char *out_bam, bool by_name, int threads, BamStats *stats);    // samtools sort [-n]
void bam_index(const char *bam, int threads);
// the alignments of a SAM or BAM file as flat arrays (what the error-profile stage counts over; ErrorProfiling.java:145-172
// reads the same fields through htsjdk): CIGAR as BAM words (len<<4|op, MIDNSHP=X), bases as BAM nibbles (=ACMGRSVTWYHKDBN)
struct AlnTable {
    std::vector<std::pair<std::string, uint32_t>> refs;       // @SQ name, length
    std::vector<int32_t> ref, pos, l_seq; std::vector<uint32_t> flag;   // pos: 0-based leftmost; ref -1: unplaced
    std::vector<uint32_t> cig_off, n_cig, cigar;
    std::vector<uint64_t> seq_off; std::vector<uint8_t> seq;   // seq_off in bases; base j of a record: nibble (seq_off + j)
    size_t n() const { return ref.size(); }
};
void load_alignments(const char *sam_or_bam, int threads, AlnTable &out);                                                           // samtools index -> <bam>.bai
}
