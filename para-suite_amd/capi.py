"""ctypes binding of libparasuite_hip.so (include/parasuite_hip.h).

The library is the product; this module only marshals arguments.  If the
shared object is missing the import fails loudly -- there is no fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("PARASUITE_LIB") or os.path.join(_HERE, "libparasuite_hip.so")   # PARASUITE_LIB: a diagnostic build of the same library


class IndexInfo(C.Structure):
    _fields_ = [("seq_len", C.c_uint64), ("l_pac", C.c_uint64), ("primary", C.c_uint64), ("L2", C.c_uint64 * 5),
                ("n_blocks", C.c_uint64), ("n_sa", C.c_uint64), ("device_bytes", C.c_uint64),
                ("n_contigs", C.c_int32), ("n_holes", C.c_int32), ("sa_rounds", C.c_int32), ("sa_intv", C.c_int32),
                ("build_ms", C.c_double), ("jump_levels", C.c_int32), ("pad_", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("ms_width", C.c_double), ("ms_backtrack", C.c_double), ("ms_compact", C.c_double),
                ("ms_select", C.c_double), ("ms_sa2pos", C.c_double), ("ms_refine", C.c_double),
                ("ms_host_post", C.c_double), ("ms_total", C.c_double), ("n_width_launches", C.c_int32),
                ("n_backtrack_launches", C.c_int32), ("n_overflow_tier1", C.c_int64), ("n_overflow_tier2", C.c_int64),
                ("ms_classify", C.c_double), ("ms_rows", C.c_double), ("ms_sel_hard", C.c_double), ("ms_sel_easy", C.c_double),
                ("bt_begin_ms", C.c_double), ("bt_end_ms", C.c_double)]


class KStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("occ_pairs", "occ_same_blk", "nodes", "pushes", "pops", "lf_steps", "iters",
                                          "exact_steps")]


ALN_DTYPE = np.dtype([("k", "<u8"), ("l", "<u8"), ("score", "<u2"), ("units", "<u2"), ("n_mm", "u1"), ("n_gapo", "u1"),
                      ("n_gape", "u1"), ("n_ins", "u1"), ("n_del", "u1"), ("pad", "u1", 7)])
HIT_DTYPE = np.dtype([("pos", "<i8"), ("sa", "<u8"), ("type", "<i4"), ("strand", "<i4"), ("mapq", "<i4"), ("n_mm", "<i4"),
                      ("n_gapo", "<i4"), ("n_gape", "<i4"), ("ref_shift", "<i4"), ("score", "<i4"), ("c1", "<i4"),
                      ("c2", "<i4"), ("n_cigar", "<i4"), ("n_multi", "<i4"), ("cigar", "<u4", 16)], align=True)

EXPORTS = ["ps_version", "ps_last_error", "ps_index", "ps_map", "ps_ctx_open", "ps_ctx_build", "ps_ctx_close",
           "ps_ctx_set_stock", "ps_ctx_set_profile", "ps_ctx_set_profile_matrix", "ps_ctx_set_tiers", "ps_ctx_set_stats", "ps_ctx_set_lanes", "ps_ctx_info",
           "ps_ctx_blob", "ps_ctx_meta", "ps_ctx_from_blobs", "ps_ctx_clone", "ps_ctx_fetch", "ps_ctx_export_blob", "ps_ctx_sa_lookup", "ps_ctx_index_check", "ps_sam_to_bam", "ps_map_to_bam", "ps_bam_view", "ps_bam_sort", "ps_bam_index", "ps_batch_from_fastq",
           "ps_batch_from_codes", "ps_batch_free", "ps_batch_n", "ps_batch_search", "ps_batch_select_hard",
           "ps_batch_select_easy", "ps_batch_locate", "ps_batch_run", "ps_batch_write_sam", "ps_batch_n_aln",
           "ps_batch_alns", "ps_batch_hits", "ps_batch_timing", "ps_batch_kstats", "ps_ctx_read_iters", "ps_parse_check", "ps_error_profile", "ps_map_profiled", "ps_release_host_cache"]

_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise ImportError("libparasuite_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(SO_PATH)
    P = C.POINTER
    L.ps_version.restype = C.c_char_p
    L.ps_last_error.restype = C.c_char_p
    L.ps_index.argtypes = [C.c_char_p]
    L.ps_map.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
    L.ps_ctx_open.argtypes = [C.c_char_p, C.c_int]
    L.ps_ctx_open.restype = C.c_void_p
    L.ps_ctx_build.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.ps_ctx_build.restype = C.c_void_p
    L.ps_ctx_close.argtypes = [C.c_void_p]
    L.ps_ctx_set_stock.argtypes = [C.c_void_p, C.c_char_p]
    L.ps_ctx_set_profile.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
    L.ps_ctx_set_profile_matrix.argtypes = [C.c_void_p, P(C.c_double), C.c_double, C.c_double, C.c_int]
    L.ps_ctx_set_tiers.argtypes = [C.c_void_p, P(C.c_uint32), P(C.c_int32), C.c_int]
    L.ps_ctx_set_stats.argtypes = [C.c_void_p, C.c_int]
    L.ps_ctx_set_lanes.argtypes = [C.c_void_p, C.c_int]
    L.ps_ctx_info.argtypes = [C.c_void_p, P(IndexInfo)]
    L.ps_ctx_blob.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_uint64)]
    L.ps_ctx_meta.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.ps_ctx_meta.restype = C.c_int64
    L.ps_ctx_from_blobs.argtypes = [C.c_char_p, C.c_int64, C.c_int, P(C.c_void_p)]
    L.ps_ctx_from_blobs.restype = C.c_void_p
    L.ps_ctx_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    L.ps_ctx_export_blob.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    L.ps_ctx_sa_lookup.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.ps_ctx_index_check.argtypes = [C.c_void_p, C.c_void_p]
    L.ps_batch_from_fastq.argtypes = [C.c_void_p, C.c_char_p]
    L.ps_batch_from_fastq.restype = C.c_void_p
    L.ps_batch_from_codes.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.ps_batch_from_codes.restype = C.c_void_p
    L.ps_batch_free.argtypes = [C.c_void_p]
    L.ps_batch_n.argtypes = [C.c_void_p]
    L.ps_batch_n.restype = C.c_int64
    L.ps_batch_search.argtypes = [C.c_void_p]
    L.ps_batch_select_hard.argtypes = [C.c_void_p, C.c_uint64, P(C.c_uint64)]
    L.ps_batch_select_easy.argtypes = [C.c_void_p, C.c_int]
    L.ps_batch_locate.argtypes = [C.c_void_p]
    L.ps_batch_run.argtypes = [C.c_void_p, C.c_int]
    L.ps_batch_write_sam.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.ps_batch_n_aln.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.ps_batch_alns.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    L.ps_batch_alns.restype = C.c_int64
    L.ps_batch_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.ps_batch_timing.argtypes = [C.c_void_p, P(Timing)]
    L.ps_ctx_read_iters.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.ps_ctx_read_iters.restype = C.c_int64
    L.ps_batch_kstats.argtypes = [C.c_void_p, C.c_int, P(KStats)]
    _LIB = L
    return L


class PsError(RuntimeError):
    pass


def _chk(rc):
    if rc:
        raise PsError(lib().ps_last_error().decode())


class Ctx:
    """Device-resident FM index + alignment options (ps_ctx)."""

    def __init__(self, handle):
        if not handle:
            raise PsError(lib().ps_last_error().decode())
        self.h = handle
        self._keep = None

    @classmethod
    def open(cls, ref_fa, device=0):
        return cls(lib().ps_ctx_open(ref_fa.encode(), device))

    @classmethod
    def build(cls, ref_fa, device=0, save_files=False):
        return cls(lib().ps_ctx_build(ref_fa.encode(), device, 1 if save_files else 0))

    @classmethod
    def from_blobs(cls, meta, device, ptrs, keep=None):
        arr = (C.c_void_p * 3)(*ptrs)
        c = cls(lib().ps_ctx_from_blobs(meta, len(meta), device, arr))
        c._keep = keep
        return c

    def clone(self, device=0):
        """a second context holding a device-to-device copy of this one's index (ps_map's route to every device after the first)"""
        lib().ps_ctx_clone.restype = C.c_void_p
        lib().ps_ctx_clone.argtypes = [C.c_void_p, C.c_int]
        return Ctx(lib().ps_ctx_clone(self.h, int(device)))

    def close(self):
        if self.h:
            lib().ps_ctx_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stock(self, n="0.04"):
        _chk(lib().ps_ctx_set_stock(self.h, str(n).encode()))

    def set_profile_files(self, ep, ip, x="-1"):
        _chk(lib().ps_ctx_set_profile(self.h, ep.encode(), ip.encode() if ip else None, str(x).encode()))

    def set_profile(self, P, ins_rate=0.0, del_rate=0.0, x=-1):
        arr = (C.c_double * 16)(*[float(v) for v in np.asarray(P, dtype=np.float64).reshape(16)])
        _chk(lib().ps_ctx_set_profile_matrix(self.h, arr, ins_rate, del_rate, int(x)))

    def set_tiers(self, pool_cap=None, aln_cap=None, bt_blocks=0):
        pc = (C.c_uint32 * 3)(*pool_cap) if pool_cap else None
        ac = (C.c_int32 * 3)(*aln_cap) if aln_cap else None
        _chk(lib().ps_ctx_set_tiers(self.h, pc, ac, bt_blocks))

    def set_stats(self, on=True):
        """search launches that follow count their Occ lookups / pushes / pops (Batch.kstats(1)); off = the timed kernel"""
        _chk(lib().ps_ctx_set_stats(self.h, 1 if on else 0))

    def set_lanes(self, n):
        """batches created afterwards take n (1 or 2) lanes of work in turn; two batches on two lanes may run from two threads"""
        _chk(lib().ps_ctx_set_lanes(self.h, int(n)))

    def info(self):
        i = IndexInfo()
        _chk(lib().ps_ctx_info(self.h, C.byref(i)))
        return i

    def blob(self, which):
        p, n = C.c_void_p(), C.c_uint64()
        _chk(lib().ps_ctx_blob(self.h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def meta(self):
        n = lib().ps_ctx_meta(self.h, None, 0)
        buf = C.create_string_buffer(n)
        lib().ps_ctx_meta(self.h, buf, n)
        return buf.raw

    def fetch(self, which):
        _, n = self.blob(which)
        out = np.empty(n, dtype=np.uint8)
        _chk(lib().ps_ctx_fetch(self.h, which, out.ctypes.data, n))
        return out

    def sa_samples(self):
        """Sampled suffix array as uint64: blob 1 holds n_sa low words followed by the bit-32 plane; entry 0 means -1."""
        n_sa = int(self.info().n_sa)
        w = self.fetch(1).view("<u4")
        out = w[:n_sa].astype(np.uint64)
        hi = (w[n_sa:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1
        out |= hi.reshape(-1)[:n_sa].astype(np.uint64) << np.uint64(32)
        out[0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        return out

    def index_check(self):
        """every row of the index against the packed text along the LF cycle -> dict(rows, bad_symbols, bad_samples, longest_arc)"""
        out = np.zeros(4, dtype=np.uint64)
        _chk(lib().ps_ctx_index_check(self.h, out.ctypes.data))
        return dict(rows=int(out[0]), bad_symbols=int(out[1]), bad_samples=int(out[2]), longest_arc=int(out[3]))

    def sa_lookup(self, rows):
        """SA[row] for rows in [1, seq_len] (device LF walk to a sampled row)"""
        r = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.empty(r.size, dtype=np.uint64)
        _chk(lib().ps_ctx_sa_lookup(self.h, r.ctypes.data, r.size, out.ctypes.data))
        return out

    def export_blob(self, which, dev_ptr, nbytes):
        _chk(lib().ps_ctx_export_blob(self.h, which, dev_ptr, nbytes))

    def bwt_syms_chunked(self, chunk_blocks=1 << 20):
        """BWT symbol string (1 byte/symbol) decoded block-wise; for genomes too big for bwt_syms()."""
        info = self.info()
        blk = self.fetch(0).view("<u4").reshape(-1, 16)
        out = np.empty(blk.shape[0] * 192, dtype=np.uint8)
        sh = np.arange(32, dtype=np.uint32)[None, None, :]
        for a in range(0, blk.shape[0], chunk_blocks):
            lo = (blk[a:a + chunk_blocks, 4:10, None] >> sh) & 1
            hi = (blk[a:a + chunk_blocks, 10:16, None] >> sh) & 1
            out[a * 192:(a + lo.shape[0]) * 192] = (lo | (hi << 1)).astype(np.uint8).reshape(-1)
        return out[:info.seq_len]

    def bwt_syms(self):
        """Decode the Occ blocks (two bit planes per block) back into the BWT symbol string; also the block counts."""
        blk = self.fetch(0).view("<u4").reshape(-1, 16)
        return self.bwt_syms_chunked(), blk[:, :4]

    RI_WORDS = 20          # ps_types.h: PS_RI_WORDS

    def read_iters(self):
        """per-read profile of the last search launch (PS_READ_ITERS=1), RI_WORDS words per read in the launch's own order of the reads:
        iterations | stack slots | D(read), D(seed) << 8, the estimate's two scans << 16 / << 24 | best score, final budget << 8, hits << 16 | 16 words of D bounds"""
        n = lib().ps_ctx_read_iters(self.h, None, 0)
        out = np.zeros(n, dtype=np.uint32)
        lib().ps_ctx_read_iters(self.h, out.ctypes.data, n)
        return out

    def batch_from_fastq(self, path):
        return Batch(lib().ps_batch_from_fastq(self.h, path.encode()), self)

    def batch_from_codes(self, codes):
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        return Batch(lib().ps_batch_from_codes(self.h, c.shape[0], c.shape[1], c.ctypes.data), self)


class Batch:
    def __init__(self, handle, ctx):
        if not handle:
            raise PsError(lib().ps_last_error().decode())
        self.h = handle
        self.ctx = ctx

    def free(self):
        if self.h:
            lib().ps_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def n(self):
        return lib().ps_batch_n(self.h)

    def search(self):
        _chk(lib().ps_batch_search(self.h))

    def select_hard(self, draws_before=0):
        out = C.c_uint64()
        _chk(lib().ps_batch_select_hard(self.h, draws_before, C.byref(out)))
        return out.value

    def select_easy(self, threads=8):
        _chk(lib().ps_batch_select_easy(self.h, threads))

    def locate(self):
        _chk(lib().ps_batch_locate(self.h))

    def run(self, threads=8):
        _chk(lib().ps_batch_run(self.h, threads))

    def write_sam(self, path, header=True, threads=8):
        _chk(lib().ps_batch_write_sam(self.h, path.encode(), 1 if header else 0, threads))

    def n_aln(self):
        out = np.zeros(self.n, dtype=np.int32)
        _chk(lib().ps_batch_n_aln(self.h, out.ctypes.data, out.size))
        return out

    def alns(self, read, cap=256):
        out = np.zeros(cap, dtype=ALN_DTYPE)
        n = lib().ps_batch_alns(self.h, read, out.ctypes.data, cap)
        if n > cap:
            return self.alns(read, n)
        return out[:n]

    def hits(self):
        out = np.zeros(self.n, dtype=HIT_DTYPE)
        _chk(lib().ps_batch_hits(self.h, out.ctypes.data, out.size))
        return out

    def timing(self):
        t = Timing()
        _chk(lib().ps_batch_timing(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def kstats(self, which):
        k = KStats()
        _chk(lib().ps_batch_kstats(self.h, which, C.byref(k)))
        return {f: getattr(k, f) for f, _ in KStats._fields_}


def ps_parse_check(path, threads=1, chunk_bytes=0):
    """host-only: (reads, bases, hash, pieces) of the read parser -- whole file, or streamed in windows as ps_map does"""
    L = lib(); L.ps_parse_check.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 4)()
    _chk(L.ps_parse_check(path.encode(), int(threads), int(chunk_bytes), out))
    return tuple(int(v) for v in out)


def ps_error_profile(mapping, ref_fa, max_read_len=101, out_prefix=None):
    """<out_prefix>.errorprofile / .indelprofile from the records of a SAM or BAM file (counted on the GPU)"""
    L = lib(); L.ps_error_profile.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p]
    _chk(L.ps_error_profile(mapping.encode(), ref_fa.encode(), int(max_read_len), out_prefix.encode() if out_prefix else None))


def ps_map_profiled(threads, mm, error_profile, indel_profile, ref_fa, reads, out_sam, min_mapq, max_read_len, profile_prefix):
    """ps_map + <profile_prefix>.errorprofile / .indelprofile of its alignments with MAPQ >= min_mapq, counted from memory"""
    enc = lambda v: v.encode() if v else None
    L = lib(); L.ps_map_profiled.argtypes = [C.c_int] + [C.c_char_p] * 6 + [C.c_int, C.c_int, C.c_char_p]
    _chk(L.ps_map_profiled(int(threads), enc(str(mm)), enc(error_profile), enc(indel_profile), enc(ref_fa), enc(reads), enc(out_sam),
                           int(min_mapq), int(max_read_len), enc(profile_prefix)))


def ps_index(ref_fa):
    _chk(lib().ps_index(ref_fa.encode()))


def ps_map(threads, mm, error_profile, indel_profile, ref_fa, fastq, out_sam):
    _chk(lib().ps_map(int(threads), str(mm).encode(), error_profile.encode() if error_profile else None,
                      indel_profile.encode() if indel_profile else None, ref_fa.encode(), fastq.encode(),
                      out_sam.encode()))


class BamStats(C.Structure):
    _fields_ = [("n_in", C.c_uint64), ("n_out", C.c_uint64), ("bam_bytes", C.c_uint64)]


def ps_sam_to_bam(sam, bam, min_mapq=0, sort_by_coordinate=False, write_index=False, threads=8):
    """SAM text -> BAM; MAPQ filter, coordinate sort and <bam>.bai in the same pass (host code, needs no GPU)."""
    L = lib()
    L.ps_sam_to_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BamStats)]
    st = BamStats()
    _chk(L.ps_sam_to_bam(sam.encode(), bam.encode(), int(min_mapq), int(bool(sort_by_coordinate)), int(bool(write_index)),
                         int(threads), C.byref(st)))
    return dict(n_in=st.n_in, n_out=st.n_out, bam_bytes=st.bam_bytes)


def ps_map_to_bam(threads, mm, error_profile, indel_profile, ref_fa, fastq, out_bam, min_mapq=0, sort_by_coordinate=False, write_index=False):
    """ps_map with the records going straight into a (MAPQ-filtered, optionally sorted + indexed) BAM: no SAM text in between"""
    L = lib()
    L.ps_map_to_bam.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(BamStats)]
    st = BamStats()
    _chk(L.ps_map_to_bam(int(threads), str(mm).encode(), error_profile.encode() if error_profile else None,
                         indel_profile.encode() if indel_profile else None, ref_fa.encode(), fastq.encode(), out_bam.encode(),
                         int(min_mapq), int(bool(sort_by_coordinate)), int(bool(write_index)), C.byref(st)))
    return dict(n_in=st.n_in, n_out=st.n_out, bam_bytes=st.bam_bytes)


def _bam_call(fn, *args):
    st = BamStats()
    _chk(fn(*args, C.byref(st)))
    return dict(n_in=st.n_in, n_out=st.n_out, bam_bytes=st.bam_bytes)


def ps_bam_view(in_bam, out_bam, min_mapq=0, threads=8):
    L = lib(); L.ps_bam_view.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(BamStats)]
    return _bam_call(L.ps_bam_view, in_bam.encode(), out_bam.encode(), int(min_mapq), int(threads))


def ps_bam_sort(in_bam, out_bam, by_name=False, threads=8):
    L = lib(); L.ps_bam_sort.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(BamStats)]
    return _bam_call(L.ps_bam_sort, in_bam.encode(), out_bam.encode(), int(bool(by_name)), int(threads))


def ps_bam_index(bam, threads=8):
    L = lib(); L.ps_bam_index.argtypes = [C.c_char_p, C.c_int]
    _chk(L.ps_bam_index(bam.encode(), int(threads)))
