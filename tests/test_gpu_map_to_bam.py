"""ps_map_to_bam: the map step and PARAsuiteMapping.java:102-152's samtools calls (view -bS, view -q; Mapping.java:85-108: sort, index)
fused -- alignment records -> BAM records -> BGZF, no SAM text in between.  Must be, record for record, what ps_sam_to_bam makes of
ps_map's SAM; both files are read back with the independent BAM reader of tests/test_bam.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _recs(path):
    from test_bam import read_bam
    text, refs, recs, _ = read_bam(path)
    for r in recs:
        del r["u0"], r["u1"]
    return text, refs, recs


def test_map_to_bam_equals_map_then_sam_to_bam(mid, workdir, monkeypatch):
    import capi
    import simulate as S
    sim = S.simulate_reads(mid["genome"], n_reads=30000, read_len=50, seed=91, indel_scale=30, n_frac=0.002, min_len=28)
    fq = os.path.join(workdir, "fuse.fq")
    S.write_fastq(fq, sim)
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    ep, ip = os.path.join(workdir, "fuse.errorprofile"), os.path.join(workdir, "fuse.indelprofile")
    with open(ep, "w") as f:
        for row in P:
            f.write("".join(repr(float(v)) + "\t" for v in row) + "\n")
    open(ip, "w").write("2.1E-5\t5.9E-4")
    fa = mid["fa"]
    if not os.path.exists(fa + ".bwt"):
        capi.ps_index(fa)
    monkeypatch.setenv("PS_CHUNK_MB", "1")                       # several pieces: the BAM is appended piece by piece
    sam = os.path.join(workdir, "fuse.sam")
    capi.ps_map(8, "-1", ep, ip, fa, fq, sam)
    n_sam = sum(1 for l in open(sam) if not l.startswith("@"))
    for tag, q, srt in (("all", 0, False), ("q10", 10, False), ("q1_sorted", 1, True)):
        two, one = os.path.join(workdir, "fuse_%s.two.bam" % tag), os.path.join(workdir, "fuse_%s.one.bam" % tag)
        st2 = capi.ps_sam_to_bam(sam, two, min_mapq=q, sort_by_coordinate=srt, write_index=srt, threads=8)
        if srt:
            monkeypatch.setenv("PS_BAM_LEVEL", "6")            # ps_sam_to_bam's level (the fused route defaults to 1): byte-identical files below
        st1 = capi.ps_map_to_bam(8, "-1", ep, ip, fa, fq, one, min_mapq=q, sort_by_coordinate=srt, write_index=srt)
        monkeypatch.delenv("PS_BAM_LEVEL", raising=False)
        assert st1["n_in"] == st2["n_in"] == n_sam == 30000 and st1["n_out"] == st2["n_out"], (tag, st1, st2)
        assert st1["bam_bytes"] == os.path.getsize(one)
        t1, r1, x1 = _recs(one)
        t2, r2, x2 = _recs(two)
        assert t1 == t2 and r1 == r2
        assert len(x1) == len(x2) == st1["n_out"]
        bad = [i for i in range(len(x1)) if x1[i] != x2[i]]
        assert not bad, (tag, len(bad), x1[bad[0]], x2[bad[0]])
        if srt:                                                  # same records in the same order through the same writer: the same bytes
            assert open(one, "rb").read() == open(two, "rb").read()
            assert open(one + ".bai", "rb").read() == open(two + ".bai", "rb").read()
        if q == 0:
            assert any(r["flag"] & 4 for r in x1) and any("D" in r["cigar"] or "I" in r["cigar"] for r in x1)
            assert any(t.startswith("XA:Z:") for r in x1 for t in r["tags"])
    # the stock first pass through the mirror class, and an input without reads
    import __graft_entry__ as ge
    m = ge.load_package().mapping.BWAMapping()
    m.executeMappingToBam(8, fa, fq, os.path.join(workdir, "fuse_stock"), 10, "2")
    _, _, xs = _recs(os.path.join(workdir, "fuse_stock.bam"))
    assert 0 < len(xs) < 30000 and all(r["mapq"] >= 10 for r in xs)
    empty = os.path.join(workdir, "fuse_none.fq")
    open(empty, "w").write("")
    st = capi.ps_map_to_bam(4, "0.04", None, None, fa, empty, os.path.join(workdir, "fuse_none.bam"))
    t0, r0, x0 = _recs(os.path.join(workdir, "fuse_none.bam"))
    assert st["n_out"] == 0 and x0 == [] and len(r0) == len(mid["genome"]) and t0.startswith("@SQ")
