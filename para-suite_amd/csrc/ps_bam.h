// ps_bam.h -- SAM text -> BAM / sorted BAM + .bai, and the same operations on BAM input (host only, zlib).
#pragma once
#include <stdint.h>
namespace ps {
struct BamStats { uint64_t n_in = 0, n_out = 0, bam_bytes = 0; };
// all throw std::runtime_error; min_mapq: records with MAPQ below it are dropped (samtools view -q)
void sam_to_bam(const char *sam_path, const char *bam_path, int min_mapq, bool sort_by_coordinate, bool write_index, int threads, BamStats *stats);
void bam_view(const char *in_bam, const char *out_bam, int min_mapq, int threads, BamStats *stats);     // samtools view -q Q -b
void bam_sort(const char *in_bam, const char *out_bam, bool by_name, int threads, BamStats *stats);    // samtools sort [-n]
void bam_index(const char *bam, int threads);                                                           // samtools index -> <bam>.bai
}
