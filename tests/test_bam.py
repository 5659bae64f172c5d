"""SAM -> BAM / sorted BAM + .bai (what PARAsuiteMapping.java:102-133 and Mapping.java:85-108 run
samtools for).  samtools is not on this machine, so the files are checked with an independent reader written here from
the SAM/BAM specification: BGZF framing, every BAM field against the SAM text, the MAPQ filter, the sort order, and
region queries through the binning + linear index against a brute-force scan.  Host code only: runs without a GPU."""
import gzip
import os
import struct
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "para-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


# ---------------------------------------------------------------- independent reader (SAMv1 §4, §5) -------
def bgzf_blocks(path):
    """[(file offset, block bytes, payload)]; checks the gzip/BC framing of every block"""
    raw = open(path, "rb").read()
    out, at = [], 0
    while at < len(raw):
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04", at
        xlen = struct.unpack_from("<H", raw, at + 10)[0]
        assert raw[at + 12:at + 14] == b"BC" and struct.unpack_from("<H", raw, at + 14)[0] == 2 and xlen == 6
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        payload = gzip.decompress(raw[at:at + bsize])
        assert len(payload) <= 0xff00 and struct.unpack_from("<I", raw, at + bsize - 4)[0] == len(payload)
        out.append((at, bsize, payload))
        at += bsize
    assert out and out[-1][2] == b"" and out[-1][1] == 28          # EOF marker block
    return out


SEQ16 = "=ACMGRSVTWYHKDBN"
CIGOPS = "MIDNSHP=X"


def read_bam(path):
    blocks = bgzf_blocks(path)
    data = b"".join(p for _, _, p in blocks)
    starts, u = [], 0                                    # uncompressed offset of every block start
    for off, _, p in blocks:
        starts.append((u, off))
        u += len(p)
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", data, at)[0]
    at += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", data, at)[0]
        name = data[at + 4:at + 4 + ln - 1].decode()
        refs.append((name, struct.unpack_from("<i", data, at + 4 + ln)[0]))
        at += 8 + ln
    recs = []
    while at < len(data):
        u0 = at
        bs, ref, pos, l_name, mapq, bin_, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiiBBHHHiiii", data, at)
        p = at + 36
        name = data[p:p + l_name - 1].decode()
        p += l_name
        cig = struct.unpack_from("<%dI" % n_cig, data, p)
        p += 4 * n_cig
        sq = data[p:p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2
        seq = "".join(SEQ16[(sq[i >> 1] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq))
        q = data[p:p + l_seq]
        p += l_seq
        qual = "*" if l_seq and q[0] == 0xff else bytes(x + 33 for x in q).decode()
        tags, end = [], at + 4 + bs
        while p < end:
            tag, ty = data[p:p + 2].decode(), chr(data[p + 2])
            p += 3
            if ty == "A":
                tags.append("%s:A:%s" % (tag, chr(data[p]))); p += 1
            elif ty in "cCsSiI":
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[ty]
                v = struct.unpack_from(fmt, data, p)[0]
                tags.append("%s:i:%d" % (tag, v)); p += struct.calcsize(fmt)
                assert ty == ("c" if -128 <= v < 0 else "s" if -32768 <= v < 0 else "i" if v < 0 else
                              "C" if v <= 255 else "S" if v <= 65535 else "I")            # smallest type, as htslib
            elif ty == "Z":
                e = data.index(b"\0", p)
                tags.append("%s:Z:%s" % (tag, data[p:e].decode())); p = e + 1
            else:
                raise AssertionError(ty)
        assert p == end
        recs.append(dict(name=name, flag=flag, ref=ref, pos=pos, mapq=mapq, bin=bin_, nref=nref, npos=npos, tlen=tlen,
                         cigar="".join("%d%s" % (c >> 4, CIGOPS[c & 15]) for c in cig) or "*", seq=seq if l_seq else "*",
                         qual=qual, tags=tags, u0=u0, u1=end))
        at = end
    return text, refs, recs, starts


def voffset_to_u(starts, v):
    coff, within = v >> 16, v & 0xffff
    for u, off in starts:
        if off == coff:
            return u + within
    raise AssertionError("virtual offset does not point at a block start")


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def reg2bins(beg, end):
    end -= 1
    out = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        out += list(range(base + (beg >> shift), base + (end >> shift) + 1))
    return out


def read_bai(path):
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\x01"
    n_ref = struct.unpack_from("<i", d, 4)[0]
    at, refs = 8, []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", d, at)[0]
        at += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", d, at)
            at += 8
            bins[b] = [struct.unpack_from("<QQ", d, at + 16 * k) for k in range(n_chunk)]
            at += 16 * n_chunk
        n_intv = struct.unpack_from("<i", d, at)[0]
        lin = list(struct.unpack_from("<%dQ" % n_intv, d, at + 4))
        at += 4 + 8 * n_intv
        refs.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", d, at)[0] if at + 8 <= len(d) else None
    return refs, n_no_coor


# ---------------------------------------------------------------- fixtures ----------------------------------
def sam_fields(path):
    head, recs = [], []
    for l in open(path):
        l = l.rstrip("\n")
        if l.startswith("@"):
            head.append(l)
        elif l:
            recs.append(l.split("\t"))
    return head, recs


@pytest.fixture(scope="module")
def sam(tmp_path_factory):
    """a real SAM of the path: the oracle's output for reads on the three-contig genome (gaps, clips, XA, unmapped)"""
    import orc
    import simulate as S
    d = tmp_path_factory.mktemp("bam")
    rng = np.random.default_rng(5)
    g = [("c%d" % i, S.make_contig(n, rng, [(50, 90)] if i == 1 else [])) for i, n in enumerate((40000, 90000, 20000))]
    g[2][1][2000:14000] = g[0][1][5000:17000]          # a second copy: reads with alternative hits (XA) and MAPQ 0
    fa = str(d / "g.fa")
    S.write_fasta(fa, g)
    sim = S.simulate_reads(g, n_reads=6000, read_len=60, min_len=30, seed=3, indel_scale=60, n_frac=0.002)
    fq = str(d / "r.fq")
    S.write_fastq(fq, sim)
    out = str(d / "r.sam")
    orc.Index.from_fasta(fa).map_fastq(orc.stock_opt("0.04"), fq, out, n_threads=4)
    return out


def _check_records(sam_recs, recs, refs):
    assert len(sam_recs) == len(recs)
    names = [r[0] for r in refs]
    for f, r in zip(sam_recs, recs):
        assert f[0] == r["name"] and int(f[1]) == r["flag"] and int(f[4]) == r["mapq"]
        assert (f[2] == "*" and r["ref"] == -1) or names[r["ref"]] == f[2]
        assert int(f[3]) - 1 == r["pos"] and f[5] == r["cigar"] and f[9] == r["seq"] and f[10] == r["qual"]
        assert r["nref"] == -1 and r["npos"] == -1 and r["tlen"] == 0
        assert f[11:] == r["tags"]
        ref_len = sum(int(n) for n, op in __import__("re").findall(r"(\d+)([MIDNSHP=X])", f[5]) if op in "MDN=X") or 1
        assert r["bin"] == reg2bin(r["pos"], r["pos"] + ref_len)      # the smallest bin that contains [pos, pos + ref_len)


def test_unsorted_bam_equals_the_sam(sam, tmp_path):
    import capi
    bam = str(tmp_path / "a.bam")
    st = capi.ps_sam_to_bam(sam, bam, threads=4)
    head, srecs = sam_fields(sam)
    text, refs, recs, _ = read_bam(bam)
    assert st["n_in"] == st["n_out"] == len(srecs) and st["bam_bytes"] == os.path.getsize(bam)
    assert text.rstrip("\n").split("\n") == head
    assert refs == [(l.split("\t")[1][3:], int(l.split("\t")[2][3:])) for l in head if l.startswith("@SQ")]
    _check_records(srecs, recs, refs)
    assert any(r["cigar"].count("I") or r["cigar"].count("D") for r in recs) and any(r["flag"] & 4 for r in recs)
    assert any(t.startswith("XA:Z:") for r in recs for t in r["tags"])


def test_mapq_filter(sam, tmp_path):
    import capi
    bam = str(tmp_path / "q.bam")
    st = capi.ps_sam_to_bam(sam, bam, min_mapq=10, threads=3)
    _, srecs = sam_fields(sam)
    keep = [f for f in srecs if int(f[4]) >= 10]
    _, refs, recs, _ = read_bam(bam)
    assert 0 < len(keep) < len(srecs) and st["n_out"] == len(keep)
    _check_records(keep, recs, refs)


def test_sorted_bam_and_index_queries(sam, tmp_path):
    import capi
    bam = str(tmp_path / "s.bam")
    capi.ps_sam_to_bam(sam, bam, min_mapq=1, sort_by_coordinate=True, write_index=True, threads=4)
    head, srecs = sam_fields(sam)
    text, refs, recs, starts = read_bam(bam)
    assert text.split("\n")[0] == "@HD\tVN:1.6\tSO:coordinate"
    key = [((r["ref"] & 0xffffffff), r["pos"]) for r in recs]
    assert key == sorted(key)                                             # coordinate order, unplaced last
    names = [r[0] for r in refs]
    keep = [f for f in srecs if int(f[4]) >= 1]
    order = sorted(range(len(keep)), key=lambda i: ((names.index(keep[i][2]) if keep[i][2] != "*" else 0xffffffff), int(keep[i][3]) - 1))
    _check_records([keep[i] for i in order], recs, refs)                  # stable: input order among equal keys
    idx, n_no_coor = read_bai(bam + ".bai")
    assert len(idx) == len(refs) and n_no_coor == sum(r["ref"] < 0 for r in recs)
    import re
    def span(r):
        n = sum(int(a) for a, op in re.findall(r"(\d+)([MIDNSHP=X])", r["cigar"]) if op in "MDN=X") or 1
        return r["pos"], r["pos"] + n
    rng = np.random.default_rng(1)
    for tid, (name, ln) in enumerate(refs):
        bins, lin = idx[tid]
        for _ in range(60):
            beg = int(rng.integers(0, ln - 1)); end = min(ln, beg + int(rng.integers(1, 40000)))
            want = [r["name"] + str(r["pos"]) for r in recs if r["ref"] == tid and span(r)[0] < end and span(r)[1] > beg]
            # query as a reader would: candidate chunks from the bins, cut by the linear index, then filter by overlap
            min_off = lin[beg >> 14] if (beg >> 14) < len(lin) else (lin[-1] if lin else 0)
            got = []
            for b in reg2bins(beg, end):
                for cb, ce in bins.get(b, []):
                    if ce <= min_off:
                        continue
                    u0, u1 = voffset_to_u(starts, cb), voffset_to_u(starts, ce)
                    got += [r["name"] + str(r["pos"]) for r in recs if u0 <= r["u0"] < u1 and r["ref"] == tid and span(r)[0] < end and span(r)[1] > beg]
            assert sorted(set(got)) == sorted(set(want)), (name, beg, end)
        meta = bins.get(37450)
        assert meta and meta[1][0] == sum(r["ref"] == tid and not r["flag"] & 4 for r in recs)


def test_errors_are_reported(tmp_path):
    import capi
    bad = tmp_path / "bad.sam"
    bad.write_text("@SQ\tSN:c0\tLN:100\nr1\t0\tnope\t1\t30\t5M\t*\t0\t0\tACGTA\tIIIII\n")
    with pytest.raises(capi.PsError):
        capi.ps_sam_to_bam(str(bad), str(tmp_path / "bad.bam"))
    with pytest.raises(capi.PsError):
        capi.ps_sam_to_bam(str(tmp_path / "missing.sam"), str(tmp_path / "x.bam"))


# ---------------------------------------------------------------- the same steps on BAM input, and the argv shim ---------
def _natural_key(name):
    """samtools' strnum_cmp: digit runs compare as numbers"""
    import re
    return [(1, int(t), 0) if t.isdigit() else (0, 0, t) for t in re.findall(r"\d+|\D", name)]


def test_steps_on_bam_input(sam, tmp_path):
    """view -q, sort, sort -n, index as separate calls on BAM files -- the sequence the unmodified Java issues"""
    import capi
    _, srecs = sam_fields(sam)
    a, q, s, n = (str(tmp_path / x) for x in ("a.bam", "q.bam", "s.bam", "n.bam"))
    capi.ps_sam_to_bam(sam, a, threads=4)
    st = capi.ps_bam_view(a, q, min_mapq=10, threads=4)
    keep = [f for f in srecs if int(f[4]) >= 10]
    text, refs, recs, _ = read_bam(q)
    assert st["n_in"] == len(srecs) and st["n_out"] == len(keep)
    _check_records(keep, recs, refs)                                          # records pass through byte for byte
    capi.ps_bam_sort(q, s, threads=4)
    text, refs, recs, starts = read_bam(s)
    assert text.split("\n")[0].endswith("SO:coordinate")
    key = [((r["ref"] & 0xffffffff), r["pos"]) for r in recs]
    assert key == sorted(key) and len(recs) == len(keep)
    # one-call and step-by-step pipelines give the same file
    one = str(tmp_path / "one.bam")
    capi.ps_sam_to_bam(sam, one, min_mapq=10, sort_by_coordinate=True, write_index=True, threads=4)
    assert read_bam(one)[2] == recs
    capi.ps_bam_index(s, threads=4)
    assert open(s + ".bai", "rb").read() == open(one + ".bai", "rb").read() or read_bai(s + ".bai")[0] == read_bai(one + ".bai")[0]
    # index of an existing file: offsets must point into THAT file
    idx, _ = read_bai(s + ".bai")
    for tid in range(len(refs)):
        for b, chunks in idx[tid][0].items():
            if b == 37450:
                continue
            for cb, ce in chunks:
                u0 = voffset_to_u(starts, cb)
                assert any(r["u0"] == u0 for r in recs)                       # every chunk starts at a record
    with pytest.raises(capi.PsError):
        capi.ps_bam_index(q)                                                  # not coordinate-sorted
    capi.ps_bam_sort(q, n, by_name=True, threads=4)
    text, _, nrecs, _ = read_bam(n)
    assert text.split("\n")[0].endswith("SO:queryname")
    names = [r["name"] for r in nrecs]
    assert names == sorted(names, key=_natural_key) and sorted(names) == sorted(f[0] for f in keep)


def test_samtools_argv_shim(sam, tmp_path):
    """para-suite_amd/bin/samtools accepts the four command lines of PARAsuiteMapping.java:102-133 / Mapping.java:85-105"""
    import subprocess
    exe = os.path.join(ROOT, "para-suite_amd", "bin", "samtools")
    assert os.path.exists(exe), "run the build first"
    p = str(tmp_path / "P")
    import shutil
    shutil.copy(sam, p + ".sam")
    run = lambda *a: subprocess.run([exe] + list(a), stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    assert run("view", "-bS", "-t", "ref.fa", p + ".sam", "-o", p + ".bam").returncode == 0
    assert run("view", "-q", "10", "-b", p + ".bam", "-o", p + ".unique.bam").returncode == 0
    assert run("sort", p + ".unique.bam", "-o", p + ".unique.bamsort.bam").returncode == 0
    os.replace(p + ".unique.bamsort.bam", p + ".unique.bam")
    assert run("index", p + ".unique.bam").returncode == 0
    _, srecs = sam_fields(sam)
    _, refs, recs, _ = read_bam(p + ".unique.bam")
    assert len(recs) == sum(int(f[4]) >= 10 for f in srecs) and os.path.getsize(p + ".unique.bam.bai") > 8
    r = run("mpileup", p + ".bam")
    assert r.returncode == 1 and b"not supported" in r.stderr
    r = run("view", "-q", "10", "-b", p + ".nope.bam", "-o", p + ".x.bam")
    assert r.returncode == 1
