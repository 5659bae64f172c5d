// ps_bam.h -- SAM text -> BAM / sorted BAM + .bai, and the same operations on BAM input (host only, zlib).
#pragma once
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>
namespace ps {
struct BamStats { uint64_t n_in = 0, n_out = 0, bam_bytes = 0; };
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
