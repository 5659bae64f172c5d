"""ctypes binding of the CPU oracle (oracle/ps_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by para-suite_amd/.  PARITY UNPINNED: see
oracle/ps_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Opt(C.Structure):
    _fields_ = [("max_diff", C.c_int32), ("fnr", C.c_double), ("max_gapo", C.c_int32), ("max_gape", C.c_int32),
                ("mode_gape", C.c_int32), ("indel_end_skip", C.c_int32), ("max_del_occ", C.c_int32),
                ("max_entries", C.c_int32), ("seed_len", C.c_int32), ("max_seed_diff", C.c_int32),
                ("max_top2", C.c_int32), ("s_mm", C.c_int32), ("s_gapo", C.c_int32), ("s_gape", C.c_int32),
                ("n_occ", C.c_int32), ("profile", C.c_int32), ("unit", C.c_int32), ("x_avg_mm", C.c_int32),
                ("sub_cost", C.c_int32 * 16), ("n_cost", C.c_int32), ("gapo_ins_cost", C.c_int32),
                ("gapo_del_cost", C.c_int32), ("gape_cost", C.c_int32)]


class Aln(C.Structure):
    _fields_ = [("k", C.c_uint64), ("l", C.c_uint64), ("n_mm", C.c_int32), ("n_gapo", C.c_int32),
                ("n_gape", C.c_int32), ("n_ins", C.c_int32), ("n_del", C.c_int32), ("score", C.c_int32),
                ("units", C.c_int32), ("_pad", C.c_int32)]


class Width(C.Structure):
    _fields_ = [("w", C.c_uint64), ("bid", C.c_int32), ("_pad", C.c_int32)]


class Hit(C.Structure):
    _fields_ = [("pos", C.c_int64), ("sa", C.c_uint64), ("type", C.c_int32), ("strand", C.c_int32),
                ("mapq", C.c_int32), ("n_mm", C.c_int32), ("n_gapo", C.c_int32), ("n_gape", C.c_int32),
                ("ref_shift", C.c_int32), ("score", C.c_int32), ("c1", C.c_int32), ("c2", C.c_int32),
                ("nm", C.c_int32), ("n_cigar", C.c_int32), ("n_multi", C.c_int32), ("flag", C.c_int32),
                ("seqid", C.c_int32), ("nn", C.c_int32), ("cigar", C.c_uint32 * 16)]


class Stats(C.Structure):
    _fields_ = [("occ_pairs", C.c_uint64), ("occ_same_blk", C.c_uint64), ("nodes", C.c_uint64),
                ("pushes", C.c_uint64), ("lf_steps", C.c_uint64), ("max_stack", C.c_uint64),
                ("top2_breaks", C.c_uint64), ("max_entries_stops", C.c_uint64)]


class Rng(C.Structure):
    _fields_ = [("x", C.c_uint64)]


def build():
    """Compile the oracle (gcc).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libps_oracle.so")
    src = os.path.join(_HERE, "ps_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        build()
    L = C.CDLL(so)
    P = C.POINTER
    L.orc_default_opt.argtypes = [P(Opt)]
    L.orc_profile_costs.argtypes = [P(Opt), P(C.c_double), C.c_double, C.c_double, C.c_int]
    L.orc_read_profile_files.argtypes = [C.c_char_p, C.c_char_p, P(C.c_double), P(C.c_double), P(C.c_double)]
    L.orc_cal_maxdiff.argtypes = [C.c_int, C.c_double, C.c_double]
    L.orc_mapq_logn.argtypes = [C.c_int]
    L.orc_srand48.argtypes = [P(Rng), C.c_long]
    L.orc_drand48.argtypes = [P(Rng)]
    L.orc_drand48.restype = C.c_double
    L.orc_lrand48.argtypes = [P(Rng)]
    L.orc_lrand48.restype = C.c_long
    L.orc_index_from_fasta.argtypes = [C.c_char_p]
    L.orc_index_from_fasta.restype = C.c_void_p
    L.orc_index_from_parts.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
    L.orc_index_from_parts.restype = C.c_void_p
    L.orc_index_free.argtypes = [C.c_void_p]
    for f in ("orc_index_seq_len", "orc_index_l_pac", "orc_index_primary", "orc_index_n_sa"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_uint64
    L.orc_index_L2.argtypes = [C.c_void_p, P(C.c_uint64)]
    L.orc_index_n_seqs.argtypes = [C.c_void_p]
    L.orc_index_n_holes.argtypes = [C.c_void_p]
    L.orc_index_pac.argtypes = [C.c_void_p]
    L.orc_index_pac.restype = C.c_void_p
    L.orc_index_bwt_syms.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_index_sa_samples.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_occ.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.orc_occ.restype = C.c_uint64
    L.orc_sa.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_sa.restype = C.c_uint64
    L.orc_set_block_syms.argtypes = [C.c_int]
    L.orc_stats_get.argtypes = [P(Stats)]
    L.orc_cal_width.argtypes = [C.c_void_p, C.c_int, C.c_void_p, P(Width)]
    L.orc_aln_one.argtypes = [C.c_void_p, P(Opt), C.c_int, C.c_void_p, P(Aln), C.c_int, P(Width), P(Width)]
    L.orc_ksw_global.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, P(C.c_uint32), C.c_int]
    L.orc_map_fastq.argtypes = [C.c_void_p, P(Opt), C.c_char_p, C.c_char_p, C.c_char_p, P(Hit), C.c_int64,
                                C.c_int, P(C.c_double), P(C.c_double)]
    L.orc_map_fastq.restype = C.c_int64
    L.orc_last_error.restype = C.c_char_p
    L.orc_set_rng_offset.argtypes = [C.c_uint64]
    L.orc_get_rng_draws.restype = C.c_uint64
    _LIB = L
    return L


def default_opt(**kw):
    o = Opt()
    lib().orc_default_opt(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def stock_opt(n="0.04", **kw):
    """`bwa aln -n <n>` semantics: a '.' in n means fnr, otherwise max_diff."""
    o = default_opt(**kw)
    if "." in str(n):
        o.fnr, o.max_diff = float(n), -1
    else:
        o.fnr, o.max_diff = -1.0, int(n)
    return o


def profile_opt(P, ins_rate=0.0, del_rate=0.0, x=-1, **kw):
    o = default_opt(**kw)
    arr = (C.c_double * 16)(*[float(v) for v in np.asarray(P, dtype=np.float64).reshape(16)])
    lib().orc_profile_costs(C.byref(o), arr, float(ins_rate), float(del_rate), int(x))
    return o


class Index:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle index: " + lib().orc_last_error().decode())
        self.h = handle

    @classmethod
    def from_fasta(cls, path):
        return cls(lib().orc_index_from_fasta(path.encode()))

    @classmethod
    def from_parts(cls, fa_path, bwt_syms, primary, sa_samples, sa_intv=32):
        b = np.ascontiguousarray(bwt_syms, dtype=np.uint8)
        s = np.ascontiguousarray(sa_samples, dtype=np.uint64)
        return cls(lib().orc_index_from_parts(fa_path.encode(), b.ctypes.data, b.size, int(primary),
                                               s.ctypes.data, int(sa_intv)))

    def __del__(self):
        try:
            lib().orc_index_free(self.h)
        except Exception:
            pass

    @property
    def seq_len(self):
        return lib().orc_index_seq_len(self.h)

    @property
    def l_pac(self):
        return lib().orc_index_l_pac(self.h)

    @property
    def primary(self):
        return lib().orc_index_primary(self.h)

    @property
    def L2(self):
        a = (C.c_uint64 * 5)()
        lib().orc_index_L2(self.h, a)
        return list(a)

    def bwt_syms(self):
        out = np.empty(self.seq_len, dtype=np.uint8)
        lib().orc_index_bwt_syms(self.h, out.ctypes.data)
        return out

    def sa_samples(self):
        out = np.empty(lib().orc_index_n_sa(self.h), dtype=np.uint64)
        lib().orc_index_sa_samples(self.h, out.ctypes.data)
        return out

    def pac(self):
        n = self.l_pac // 4 + 1
        return np.ctypeslib.as_array(C.cast(lib().orc_index_pac(self.h), C.POINTER(C.c_uint8)), shape=(n,)).copy()

    def forward_codes(self):
        p = self.pac()
        i = np.arange(self.l_pac, dtype=np.int64)
        return (p[i >> 2] >> ((~i & 3) << 1)) & 3

    def occ(self, k, c):
        return lib().orc_occ(self.h, int(k), int(c))

    def sa(self, k):
        return lib().orc_sa(self.h, int(k))

    def cal_width(self, rev_read):
        r = np.ascontiguousarray(rev_read, dtype=np.uint8)
        w = (Width * (r.size + 1))()
        lib().orc_cal_width(self.h, r.size, r.ctypes.data, w)
        return [(x.w, x.bid) for x in w]

    def aln_one(self, opt, read, cap=256, want_width=False):
        r = np.ascontiguousarray(read, dtype=np.uint8)
        out = (Aln * cap)()
        w = (Width * (r.size + 1))()
        sw = (Width * (opt.seed_len + 1))()
        n = lib().orc_aln_one(self.h, C.byref(opt), r.size, r.ctypes.data, out, cap, w, sw)
        alns = [dict(k=a.k, l=a.l, n_mm=a.n_mm, n_gapo=a.n_gapo, n_gape=a.n_gape, n_ins=a.n_ins, n_del=a.n_del,
                     score=a.score, units=a.units) for a in out[:min(n, cap)]]
        if want_width:
            return n, alns, [(x.w, x.bid) for x in w], [(x.w, x.bid) for x in sw]
        return n, alns

    def map_fastq(self, opt, fastq, sam_out, sai_out=None, n_threads=1, want_hits=0, draws_before=0):
        hits = (Hit * want_hits)() if want_hits else None
        lib().orc_set_rng_offset(int(draws_before))
        ta, ts = C.c_double(), C.c_double()
        n = lib().orc_map_fastq(self.h, C.byref(opt), fastq.encode(), sam_out.encode(),
                                sai_out.encode() if sai_out else None, hits, want_hits, n_threads,
                                C.byref(ta), C.byref(ts))
        if n < 0:
            raise RuntimeError("oracle map: " + lib().orc_last_error().decode())
        lib().orc_set_rng_offset(0)
        tt = (C.c_double * 4)()
        lib().orc_last_times(tt)
        return dict(n=n, t_aln=ta.value, t_samse=ts.value, t_parse=tt[0], t_samse_records=tt[2], t_sam_text=tt[3],
                    hits=hits, draws_after=lib().orc_get_rng_draws())


def error_profile(sam_path, fasta_path, max_read_len, out_prefix):
    """CPU restatement of ErrorProfiling.inferErrorProfile (oracle/orc_profile.c): writes <out_prefix>.errorprofile and
    .indelprofile, returns the number of records processed"""
    L = lib()
    L.orc_error_profile.restype = C.c_long
    L.orc_error_profile.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p]
    L.orc_profile_last_error.restype = C.c_char_p
    n = L.orc_error_profile(sam_path.encode(), fasta_path.encode(), int(max_read_len), out_prefix.encode())
    if n < 0:
        raise RuntimeError("oracle error profile: " + L.orc_profile_last_error().decode())
    return n


def java_double(v):
    """java.lang.Double.toString as the oracle prints it"""
    L = lib()
    L.orc_java_double.argtypes = [C.c_double, C.c_char_p]
    buf = C.create_string_buffer(64)
    L.orc_java_double(float(v), buf)
    return buf.value.decode()


def ksw_global(query, target, w):
    q = np.ascontiguousarray(query, dtype=np.uint8)
    t = np.ascontiguousarray(target, dtype=np.uint8)
    cig = (C.c_uint32 * 64)()
    n = lib().orc_ksw_global(q.size, q.ctypes.data, t.size, t.ctypes.data, int(w), cig, 64)
    return [(c >> 4, "MIDS"[c & 0xF]) for c in cig[:n]]


def stats(reset=False):
    s = Stats()
    lib().orc_stats_get(C.byref(s))
    d = {k: getattr(s, k) for k, _ in Stats._fields_}
    if reset:
        lib().orc_stats_reset()
    return d


def read_sai(path):
    """Parse the oracle's sai dump: per read int32 n_aln + n_aln * Aln."""
    out = []
    raw = open(path, "rb").read()
    off = 0
    while off < len(raw):
        n = int.from_bytes(raw[off:off + 4], "little", signed=True)
        off += 4
        a = np.frombuffer(raw, dtype=np.dtype([("k", "<u8"), ("l", "<u8"), ("n_mm", "<i4"), ("n_gapo", "<i4"),
                                               ("n_gape", "<i4"), ("n_ins", "<i4"), ("n_del", "<i4"),
                                               ("score", "<i4"), ("units", "<i4"), ("pad", "<i4")]),
                          count=n, offset=off)
        off += n * 48
        out.append(a)
    return out
