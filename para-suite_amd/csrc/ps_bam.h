// ps_bam.h -- SAM text -> BAM / sorted BAM + .bai (host only, zlib).
#pragma once
#include <stdint.h>
namespace ps {
struct BamStats { uint64_t n_in = 0, n_out = 0, bam_bytes = 0; };
// throws std::runtime_error; min_mapq: records with MAPQ below it are dropped (samtools view -q)
void sam_to_bam(const char *sam_path, const char *bam_path, int min_mapq, bool sort_by_coordinate, bool write_index, int threads, BamStats *stats);
}
