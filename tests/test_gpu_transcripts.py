"""GPU parity on the shape of the reference's TRANSCRIPT pass (Main.java:392-416: the reads go against a FASTA of transcripts,
MAPQ >= 1, Main.java:115): many short reference sequences with `Gene|Transcript|Chr|starts;...|ends;...|strand` headers (the
format of examples/references/reference_chr1_transcripts.fa:1 -- generated here, nothing is copied from there), adapter-trimmed
PAR-CLIP read lengths (18-40 bp), a good share of the reads cut ACROSS the end of one sequence into the next one.

What upstream does there, and what must therefore come out identically on both sides: the index is ONE text (all sequences
concatenated, forward + reverse complement), so a read can hit a position whose span leaves its sequence -- `bwa samse` then prints
the record with flag 4 (unmapped) but keeps RNAME/POS (bwa_print_sam1's "bridges two adjacent reference sequences" rule); @SQ lines
follow the FASTA order; RNAMEs are the header up to the first blank, `|` and `;` included.  Oracle = this repository's CPU
restatement (parity unpinned, oracle/ps_oracle.h)."""
import os

import numpy as np
import pytest

from conftest import sam_records, sam_sq

pytestmark = pytest.mark.gpu

N_CONTIGS = 96


@pytest.fixture(scope="module")
def transcripts(workdir):
    import orc
    import simulate as S
    rng = np.random.default_rng(0x7A5C)
    contigs = []
    at = 10000
    for i in range(N_CONTIGS):
        n = int(rng.integers(150, 1601))
        exons = sorted(rng.integers(at, at + 5000, size=2 * int(rng.integers(1, 5))).tolist())
        name = "GENE%d.%d|TR%05d|chr%d|%s|%s|%s" % (i // 3, i % 3, i, 1 + i % 3, ";".join(str(x) for x in exons[0::2]),
                                                    ";".join(str(x) for x in exons[1::2]), "+-"[i & 1])
        asc = S.make_contig(n, rng, [], softmask_frac=0.2)
        contigs.append((name, asc))
        at += 7000
    # isoforms share exons: copies between transcripts -> multi-mapping reads (MAPQ 0, XA) across sequences
    for i in range(0, N_CONTIGS - 1, 6):
        a, b = contigs[i][1], contigs[i + 1][1]
        n = min(a.size, b.size, 400) - 20
        b[10:10 + n] = a[5:5 + n]
    contigs[7][1][40:52] = ord("N")                       # a hole inside a short sequence
    fa = os.path.join(workdir, "transcripts.fa")
    with open(fa, "wb") as f:
        for name, asc in contigs:
            f.write(b">" + name.encode() + b" some description text\n")
            for i in range(0, asc.size, 60):
                f.write(asc[i:i + 60].tobytes() + b"\n")
    # reads of 18-40 bp cut from the CONCATENATION: about a third straddle the end of a sequence
    codes_all = np.concatenate([S.contig_codes(a) for _, a in contigs])
    ends = np.cumsum([a.size for _, a in contigs])
    n_reads, Lmax = 20000, 40
    lens = rng.integers(18, Lmax + 1, size=n_reads).astype(np.int32)
    start = np.empty(n_reads, dtype=np.int64)
    bridge = rng.random(n_reads) < 0.33
    e_pick = ends[rng.integers(0, N_CONTIGS - 1, size=n_reads)]
    start[bridge] = e_pick[bridge] - rng.integers(1, lens[bridge])                # 1 .. len-1 bases before the end
    start[~bridge] = rng.integers(0, codes_all.size - Lmax, size=int((~bridge).sum()))
    start = np.clip(start, 0, codes_all.size - Lmax)
    codes = np.full((n_reads, Lmax), 255, dtype=np.uint8)
    P = S.EXAMPLE_PROFILE
    cdf = np.cumsum(P, axis=1)
    for r in range(n_reads):
        s = codes_all[start[r]:start[r] + lens[r]].copy()
        s[s > 3] = 0
        if rng.random() < 0.5:
            s = (3 - s)[::-1]
        if rng.random() < 0.6:                                                   # T->C conversions on bound reads
            ts = np.nonzero(s == 3)[0]
            if ts.size:
                s[rng.choice(ts, size=min(ts.size, int(rng.integers(1, 4))), replace=False)] = 1
        u = rng.random(s.size)
        s = (u[:, None] > cdf[s]).sum(1).astype(np.uint8)                        # sequencing errors by the profile row
        if rng.random() < 0.004:
            s[int(rng.integers(0, s.size))] = 4
        codes[r, :lens[r]] = s
    sim = dict(codes=codes, lens=lens, quals=np.full((n_reads, Lmax), 70, dtype=np.uint8))
    fq = os.path.join(workdir, "transcripts.fq")
    S.write_fastq(fq, sim, names=["SEQ_ID:%s:%d" % (contigs[int(np.searchsorted(ends, start[r], side="right"))][0], r) for r in range(n_reads)])
    return dict(fa=fa, fq=fq, contigs=contigs, n_reads=n_reads, bridge=bridge, orc_index=orc.Index.from_fasta(fa))


def _check(ctx, tr, opt, workdir, tag):
    b = ctx.batch_from_fastq(tr["fq"])
    b.run(threads=4)
    gsam, osam = os.path.join(workdir, tag + ".gpu.sam"), os.path.join(workdir, tag + ".orc.sam")
    b.write_sam(gsam)
    tr["orc_index"].map_fastq(opt, tr["fq"], osam, n_threads=8)
    sq = sam_sq(gsam)
    assert sq == sam_sq(osam) and len(sq) == N_CONTIGS
    assert [l.split("\t")[1][3:] for l in sq] == [n for n, _ in tr["contigs"]]           # FASTA order, `|` and `;` kept, description dropped
    assert [int(l.split("\t")[2][3:]) for l in sq] == [a.size for _, a in tr["contigs"]]
    g, o = sam_records(gsam), sam_records(osam)
    assert len(g) == len(o) == tr["n_reads"]
    bad = [i for i in range(len(g)) if g[i] != o[i]]
    assert not bad, (tag, len(bad), g[bad[0]], o[bad[0]])
    return gsam, g


def test_transcript_pass_shape(transcripts, workdir):
    """stock `aln -n 2` (BWAMapping.java:51-61, the first pass) and the profile pass (PARAsuiteMapping.java:63-77) against the
    transcript-shaped reference: SAM identical to the oracle's, then the MAPQ >= 1 filter of the transcript pass through the BAM
    writer, read back with the independent reader of tests/test_bam.py"""
    import capi
    import orc
    import simulate as S
    from test_bam import read_bam, sam_fields, _check_records
    tr = transcripts
    ctx = capi.Ctx.build(tr["fa"])
    assert ctx.info().n_contigs == N_CONTIGS
    ctx.set_stock("2")
    gsam, g = _check(ctx, tr, orc.stock_opt("2"), workdir, "tr_stock")
    f = [l.split("\t") for l in g]
    flags = np.array([int(x[1]) for x in f])
    # the shape is really exercised: reads that bridge two sequences are printed unmapped WITH a position, reads inside the
    # shared exons are repeats with alternative hits on another sequence
    bridged = [x for x in f if int(x[1]) & 4 and x[2] != "*"]
    assert len(bridged) > 200, len(bridged)
    assert all(x[5] != "*" and int(x[3]) > 0 for x in bridged)
    assert sum(1 for x in f if not int(x[1]) & 4) > 8000
    assert sum(1 for x in f if any(t.startswith("XA:Z:") for t in x[11:])) > 100
    assert any("|" in x[2] and ";" in x[2] for x in f if x[2] != "*")
    assert ((flags & 16) != 0).sum() > 2000
    # the profile pass on the same reads
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    ctx.set_profile(P, 2.1e-5, 5.9e-4, -1)
    psam, _ = _check(ctx, tr, orc.profile_opt(P, 2.1e-5, 5.9e-4, -1), workdir, "tr_profile")
    # MAPQ >= 1 (Main.java:115) into a BAM, as PARAsuiteMapping.java:102-133 does with samtools
    bam = os.path.join(workdir, "tr_profile.q1.bam")
    st = capi.ps_sam_to_bam(psam, bam, min_mapq=1, threads=4)
    head, srecs = sam_fields(psam)
    keep = [x for x in srecs if int(x[4]) >= 1]
    text, refs, recs, _ = read_bam(bam)
    assert 0 < len(keep) < len(srecs) and st["n_out"] == len(keep)
    assert [r[0] for r in refs] == [n for n, _ in tr["contigs"]]
    _check_records(keep, recs, refs)
