/*
 * parasuite_hip.h -- C ABI of libparasuite_hip.so, the MI355X (gfx950) replacement for the
 * aligner child processes of PARA-suite's `map` command.
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   ps_index        <- exec `bwa index <ref>`                     src/src/mapping/PARAsuiteMapping.java:45-55,
 *                                                                 src/src/mapping/BWAMapping.java:35-45
 *   ps_map          <- exec `bwa parasuite -t T -X mm -p EP -g IP ref fq -f P.sai`
 *                      + `bwa samse ref P.sai fq -f P.sam`        PARAsuiteMapping.java:63-77,84-92
 *                      (error_profile == NULL: `bwa aln -t T -n mm` + samse,  BWAMapping.java:51-61,68-75)
 *   ps_last_error   <- child exit status + inherited stderr       src/src/mapping/Mapping.java:151-198
 * The Java method these sit behind is Mapping.executeMapping(int threads, String reference,
 * String input, String outputPrefix, int mappingQualityFilter, String additionalOptions)
 * (Mapping.java:40-42); INTEGRATION.md shows the JNI subclass and the argv-compatible `bwa` shim.
 *
 * Conventions: UTF-8 paths, int return 0 = success / non-zero = failure (message via
 * ps_last_error() and on stderr), synchronous, one call at a time per process, never exit()s or
 * throws across the boundary.  There is no CPU implementation: without a HIP device every
 * computing entry point fails.
 */
#ifndef PARASUITE_HIP_H
#define PARASUITE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char *ps_version(void);
const char *ps_last_error(void);

/* `bwa index`: writes <ref_fa>.bwt/.sa/.pac/.ann (this project's formats; the caller only tests that
 * <ref_fa>.bwt exists, PARAsuiteMapping.java:45-46).  Suffix sorting runs on the GPU. */
int ps_index(const char *ref_fa);

/* `bwa parasuite` (or `bwa aln` when error_profile is NULL) + `bwa samse` fused: FASTQ in, SAM out.
 * threads: host worker threads for parsing/formatting (the reference's -t).  mm: the -X (profile
 * mode) or -n (stock mode) argument as the Java passes it, e.g. "-1", "2", "0.04".
 * Devices: the first PARASUITE_GPUS (environment, default 1), each with its own copy of the index; the
 * input is cut into pieces that a parser thread, the device workers and a SAM writer work on side by
 * side; the output does not depend on the cut or on the number of devices. */
int ps_map(int threads, const char *mm, const char *error_profile, const char *indel_profile,
           const char *ref_fa, const char *fastq, const char *out_sam);

/* ---- staged interface (same code path as ps_map; used by the argv shim, tests and bench.py) ---- */
typedef struct ps_ctx ps_ctx;
typedef struct ps_batch ps_batch;

ps_ctx *ps_ctx_open(const char *ref_fa, int device);                 /* load <ref_fa>.bwt ... into HBM  */
ps_ctx *ps_ctx_clone(ps_ctx *src, int device);                  /* a context on `device` with a device-to-device copy of src's index (what ps_map gives every device after the first) */
ps_ctx *ps_ctx_build(const char *ref_fa, int device, int save_files);/* build the index (GPU), keep it resident */
void    ps_ctx_close(ps_ctx *);
int     ps_ctx_set_stock(ps_ctx *, const char *n_arg);               /* bwa aln -n */
int     ps_ctx_set_profile(ps_ctx *, const char *error_profile, const char *indel_profile, const char *x_arg);
int     ps_ctx_set_profile_matrix(ps_ctx *, const double P[16], double ins_rate, double del_rate, int x);
int     ps_ctx_set_tiers(ps_ctx *, const uint32_t pool_cap[3], const int32_t aln_cap[3], int bt_blocks);
/* measurement runs: the search kernel of the following launches counts its search steps, pushes, pops ... and walks the plain Occ
 * blocks (no jump table): the counts are those of the reference algorithm (ps_batch_kstats, which = 1).  The default kernel
 * keeps two counters only -- occ_pairs / occ_same_blk of the steps IT takes through the Occ array (the rest go through the
 * jump table) -- and leaves the others zero */
int     ps_ctx_set_stats(ps_ctx *, int on);
/* lanes of work (stream + workspace) the batches created afterwards take in turn: 1 (default) or 2 -- two batches on two lanes may
 * be driven from two threads at once, one's selection / SA walk / DP stages then run under the other's search kernel */
int     ps_ctx_set_lanes(ps_ctx *, int n);

typedef struct {                       /* index geometry + build facts */
    uint64_t seq_len, l_pac, primary, L2[5], n_blocks, n_sa, device_bytes;
    int32_t n_contigs, n_holes, sa_rounds, sa_intv;
    double build_ms;
    int32_t  jump_levels, pad_;        /* levels of the jump table derived from the index on the device (0: none) */
} ps_index_info;
int     ps_ctx_info(ps_ctx *, ps_index_info *out);
/* index blobs (0 Occ blocks, 1 sampled SA, 2 pac) for the one-off RCCL broadcast over xGMI */
int     ps_ctx_blob(ps_ctx *, int which, void **dev_ptr, uint64_t *bytes);
int64_t ps_ctx_meta(ps_ctx *, char *buf, int64_t cap);               /* serialised contig/hole table + scalars */
ps_ctx *ps_ctx_from_blobs(const char *meta, int64_t meta_len, int device, void *const dev_ptrs[3]); /* borrowed device memory */
int     ps_ctx_fetch(ps_ctx *, int which, void *host_dst, uint64_t bytes); /* D2H copy of a blob (tests) */
int     ps_ctx_export_blob(ps_ctx *, int which, void *dev_dst, uint64_t bytes); /* D2D copy into caller memory (broadcast source) */
int     ps_ctx_index_check(ps_ctx *, uint64_t out[4]);   /* every row of the index checked against the packed text along the LF cycle: rows visited (== seq_len + 1), BWT symbol mismatches, SA sample mismatches, longest arc */
int     ps_ctx_sa_lookup(ps_ctx *, const uint64_t *rows, int64_t n, uint64_t *out); /* SA[row], rows in [1, seq_len]: index checks from the text */

ps_batch *ps_batch_from_fastq(ps_ctx *, const char *fastq);
ps_batch *ps_batch_from_codes(ps_ctx *, int64_t n, int len, const uint8_t *codes); /* [n][len], 0..3 ACGT, 4 N */
void      ps_batch_free(ps_batch *);
int64_t   ps_batch_n(ps_batch *);
int       ps_batch_search(ps_batch *);                               /* width + backtracking kernels */
int       ps_batch_select_hard(ps_batch *, uint64_t draws_before, uint64_t *draws_after);
int       ps_batch_select_easy(ps_batch *, int threads);
int       ps_batch_locate(ps_batch *);                               /* SA walk + MAPQ + banded DP kernels */
int       ps_batch_run(ps_batch *, int threads);                     /* the four stages above, single process */
int       ps_batch_write_sam(ps_batch *, const char *path, int with_header, int threads);

typedef struct { uint64_t k, l; uint16_t score, units; uint8_t n_mm, n_gapo, n_gape, n_ins, n_del, pad[7]; } ps_aln;   /* 32 B */
typedef struct {
    int64_t pos; uint64_t sa; int32_t type, strand, mapq, n_mm, n_gapo, n_gape, ref_shift, score, c1, c2,
            n_cigar, n_multi; uint32_t cigar[16];
} ps_hit;
typedef struct {
    double ms_width, ms_backtrack, ms_compact, ms_select, ms_sa2pos, ms_refine, ms_host_post, ms_total;
    int32_t n_width_launches, n_backtrack_launches;
    int64_t n_overflow_tier1, n_overflow_tier2;
    double ms_classify, ms_rows, ms_sel_hard, ms_sel_easy;   /* host sub-stages */
    double bt_begin_ms, bt_end_ms;   /* the search launches of this batch on the context's clock: two batches on two lanes overlap */
} ps_timing;
typedef struct { uint64_t occ_pairs, occ_same_blk, nodes, pushes, pops, lf_steps, iters, exact_steps; } ps_kstats;

int     ps_batch_n_aln(ps_batch *, int32_t *out, int64_t cap);       /* per read, input order */
int64_t ps_batch_alns(ps_batch *, int64_t read, ps_aln *out, int64_t cap);
int     ps_batch_hits(ps_batch *, ps_hit *out, int64_t cap);
int     ps_batch_timing(ps_batch *, ps_timing *out);
int64_t ps_ctx_read_iters(ps_ctx *, uint32_t *out, int64_t cap);   /* profiling aid (env PS_READ_ITERS=1): per read of the last search launch 20 words -- iterations | stack slots used | lower bounds as fetched (read, seed << 8) and the effort estimate's two scans (<< 16, << 24) | best score, final budget << 8, hits << 16 | 16 words of D bounds -- in the order the launch held the reads (leading-base order of the bin, not input order); returns the word count */
int     ps_batch_kstats(ps_batch *, int which /*0 width 1 backtrack 2 sa2pos*/, ps_kstats *out);
/* host-only check of the read parser: whole file on `threads` threads (chunk_bytes 0) or streamed in windows of chunk_bytes as
 * ps_map does; out = {reads, bases, order-sensitive hash of names / sequences / qualities, pieces} */
/* the library keeps up to 3 GB of page-locked host buffers between calls (locking and unlocking them costs ~0.1 s per piece of a
 * ps_map call); this gives them back */
void    ps_release_host_cache(void);
int     ps_parse_check(const char *reads_path, int threads, uint64_t chunk_bytes, uint64_t out[4]);

/* ---- after the map step --------------------------------------------------------------
 * One call for what PARAsuiteMapping.java:102-133 (`samtools view -bS -t ref x.sam -o x.bam`, `samtools view -q <mapq> -b`)
 * and Mapping.java:85-108 (`samtools sort`, `samtools index`) spawn four processes for: SAM text -> BAM, records with
 * MAPQ < min_mapq dropped, optionally sorted by coordinate (unplaced reads last, input order kept among equals) with
 * <bam>.bai next to it.  Host code (zlib); needs no GPU. */
typedef struct { uint64_t n_in, n_out, bam_bytes; } ps_bam_stats;
int     ps_sam_to_bam(const char *sam, const char *bam, int min_mapq, int sort_by_coordinate, int write_index, int threads,
                      ps_bam_stats *stats /* may be NULL */);
/* ps_map and the steps above in ONE call, without the SAM text in between (PARAsuiteMapping.java:63-152 end to end: `bwa parasuite|aln`,
 * `bwa samse`, `samtools view -bS`, `view -q`, and Mapping.java:85-108's sort + index): alignment records -> BAM records -> BGZF, piece by
 * piece while later pieces of the input are searched.  out_bam holds the records with MAPQ >= min_mapq in input order, or sorted by
 * coordinate (+ <out_bam>.bai with write_index).  Record for record what ps_sam_to_bam(ps_map(...)) writes. */
int     ps_map_to_bam(int threads, const char *mm, const char *error_profile, const char *indel_profile, const char *ref_fa, const char *fastq,
                      const char *out_bam, int min_mapq, int sort_by_coordinate, int write_index, ps_bam_stats *stats /* may be NULL */);
/* the same steps one by one on BAM input, as the unmodified Java issues them (an argv-compatible `samtools` for exactly
 * these four command shapes is built as para-suite_amd/bin/samtools):
 *   samtools view -q <mapq> -b in.bam -o out.bam   (PARAsuiteMapping.java:121-133)
 *   samtools sort [-n] in.bam -o out.bam            (Mapping.java:86-93, 118-126; -n: names ordered as samtools does)
 *   samtools index in.bam                           (Mapping.java:100-105)  -> in.bam.bai, the BAM is not rewritten */
int     ps_bam_view(const char *in_bam, const char *out_bam, int min_mapq, int threads, ps_bam_stats *stats);
int     ps_bam_sort(const char *in_bam, const char *out_bam, int by_name, int threads, ps_bam_stats *stats);
int     ps_bam_index(const char *bam, int threads);

/* ---- between the two mapping passes -------------------------------------------------------
 * What `new ErrorProfiling(mapping, reference, maxReadLength).inferErrorProfile(false, false)` writes for the second pass
 * (src/src/utils/errorprofile/ErrorProfiling.java:100-631, called at src/src/main/Main.java:327-334): counted on the GPU over
 * the records of the first pass's SAM or BAM file (unmapped, duplicate and position-less records skipped, :155-166; counts
 * in read orientation, :301-306,376-377), against the reference of an existing index (<ref_fa>.pac/.ann):
 *   <out_prefix>.errorprofile  4 lines x 4 values, row = reference base, each Double.toString + TAB (NaN if never seen), :504-531
 *   <out_prefix>.indelprofile  "<ins>\t<del>" without newline, :545-591
 * out_prefix NULL or "": the mapping file's name, as in the Java.  A read longer than max_read_len is an error (the
 * Java's arrays would overflow). */
int     ps_error_profile(const char *mapping_sam_or_bam, const char *ref_fa, int max_read_len, const char *out_prefix);
/* The first pass and its error profile in one call -- "fed directly from alignment results" (SURVEY.md §8f rank 4): ps_map, and
 * while the SAM is written the same records, straight from memory, go through the counting kernel: the alignments with
 * MAPQ >= min_mapq, i.e. what the pass's filtered BAM holds (samtools view -q, PARAsuiteMapping.java:124-133 /
 * BWAMapping.java) and ErrorProfiling would read back (Main.java:327-334).  Writes <profile_prefix>.errorprofile and
 * .indelprofile, the same bytes as ps_error_profile on that filtered file.  No BAM round trip between the passes. */
int     ps_map_profiled(int threads, const char *mm, const char *error_profile, const char *indel_profile,
                        const char *ref_fa, const char *reads_fq, const char *out_sam,
                        int min_mapq, int max_read_len, const char *profile_prefix);

#ifdef __cplusplus
}
#endif
#endif
