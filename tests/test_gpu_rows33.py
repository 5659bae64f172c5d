"""GPU parity above 2^32 BWT rows: 33-bit rows through the index builder, the search stack, the SA walk and SAM.

A 2.2 Gbp synthetic genome (4.4e9 rows > 2^32) is the smallest input that exercises the high bit everywhere;
the oracle adopts the product's BWT (it has its own Occ code, search, tie-break, locate and SAM writer) and maps
the same reads on the CPU -- after that BWT, the sampled SA and the C array have been checked against the TEXT: suffix order of
123,000 adjacent row pairs (random, and across the 2^32 boundary), BWT[r] == T[SA[r]-1], L2 == base counts.  Besides equality with the oracle the test checks a size-independent property: every
exact read maps back to the position it was cut from, on either strand, including positions above 2^32.
Takes about two minutes on one MI355X; everything (genome, reads) is generated on the device as in bench.py."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "para-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


def _strip(path):                 # QNAME and QUAL dropped: the device batch is built from codes
    out = []
    for l in open(path):
        if not l.startswith("@"):
            f = l.rstrip("\n").split("\t")
            out.append("\t".join(f[1:10] + f[11:]))
    return out


def test_parity_and_round_trip_above_2_pow_32_rows(tmp_path):
    import torch
    import bench
    import capi
    import orc
    import simulate as S

    dev = torch.device("cuda", 0)
    mbp, n_reads, L = 2200, 30000, 50
    contigs = bench.gen_genome(torch, dev, mbp * 1_000_000, 6, 0x5EED0102)
    fa = str(tmp_path / "g.fa")
    bench.write_fasta(fa, contigs)
    torch.cuda.empty_cache()
    ctx = capi.Ctx.build(fa, device=0)
    info = ctx.info()
    assert info.seq_len == 2 * mbp * 1_000_000 and info.seq_len > 2 ** 32
    assert sum(int(info.L2[c + 1]) - int(info.L2[c]) for c in range(4)) == info.seq_len

    # ---- the index checked WITHOUT trusting it: properties computed from the text (tests/index_props.py; the same check fails
    # on a suffix array with unsorted ties, swapped neighbours or a wrong position bit: tests/test_index_props_cpu.py).
    # The text: the packed forward strand, itself compared with the FASTA codes at every non-N position.
    sys.path.insert(0, HERE)
    import index_props as IP
    pac = ctx.fetch(2)
    fwd = np.empty(pac.size * 4, dtype=np.uint8)
    for k in range(4):
        fwd[k::4] = (pac >> (6 - 2 * k)) & 3
    fwd = fwd[:info.l_pac]
    at = 0
    for _, c in contigs:
        h = c.cpu().numpy()
        keep = h < 4
        assert np.array_equal(fwd[at:at + h.size][keep], h[keep]), "pac differs from the FASTA"
        at += h.size
    assert at == info.l_pac
    text = IP.Text(fwd)
    rng = np.random.default_rng(5)
    rows = np.concatenate([rng.integers(1, info.seq_len - 1, 120_000), np.arange(2 ** 32 - 1500, 2 ** 32 + 1500)]).astype(np.int64)
    bwt_all = ctx.bwt_syms_chunked()
    primary = int(info.primary)

    def bwt_lookup(r):
        r = np.asarray(r, dtype=np.int64)
        out = bwt_all[np.minimum(r - (r >= primary), bwt_all.size - 1)].astype(np.int16)
        out[r == primary] = 255
        return out
    seen = IP.check_index(text, rows, lambda r: ctx.sa_lookup(r), bwt_lookup, primary, [int(info.L2[c]) for c in range(5)])
    assert seen["pairs"] == rows.size and seen["above_2_32"] > 1000
    assert seen["long_lcp"] > 1000 and seen["max_lcp"] > 500       # pairs inside the 2 kb two-copy segments: what only the late doubling rounds order
    P = np.array(bench.PROFILE)
    P[3, 1], P[3, 3] = 0.12, 0.87

    # ---- size-independent property: exact reads come home, both strands, positions beyond 2^32 on the reverse strand
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    name, codes = contigs[-1]                      # last contig: its reverse-strand rows are the lowest, its forward the highest
    starts = (torch.rand(4000, generator=g, device=dev, dtype=torch.float64) * (codes.numel() - L - 1)).long()
    ex = codes[starts[:, None] + torch.arange(L, device=dev)[None, :]]
    keep = (ex < 4).all(1)
    ex, starts = ex[keep].cpu().numpy(), starts[keep].cpu().numpy()
    rc = (3 - ex[1::2, ::-1]).copy()               # every second read reverse-complemented
    ex[1::2] = rc
    ctx.set_stock("0")
    b = ctx.batch_from_codes(np.ascontiguousarray(ex))
    b.run(8)
    hits = b.hits()
    off = sum(c.numel() for _, c in contigs[:-1])
    ok = (hits["type"] != 0) & (hits["pos"] == off + starts)
    uniq = hits["type"] == 1
    assert uniq.mean() > 0.9
    assert ok[uniq].all()
    assert (hits["strand"] == (np.arange(ex.shape[0]) % 2))[uniq].all()      # flag 16 exactly for the reverse-complemented reads
    b.free()

    # ---- equality with the CPU oracle on simulated PAR-CLIP reads (profile costs, gapped extension)
    ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1)
    rd = bench.gen_reads(torch, dev, contigs, n_reads, L, 0x5EED0103, indels=True)
    del contigs
    torch.cuda.empty_cache()
    sim = dict(codes=rd, lens=np.full(n_reads, L, dtype=np.int32), quals=np.full((n_reads, L), 73, dtype=np.uint8))
    fq = str(tmp_path / "r.fq")
    S.write_fastq(fq, sim, names=["r%d" % i for i in range(n_reads)])
    oix = orc.Index.from_parts(fa, bwt_all, info.primary, ctx.sa_samples())     # adopted -- and checked against the text above
    osam, gsam = str(tmp_path / "o.sam"), str(tmp_path / "g.sam")
    oix.map_fastq(orc.profile_opt(P, bench.INS_RATE, bench.DEL_RATE, -1), fq, osam, n_threads=16)
    b = ctx.batch_from_codes(rd)
    b.run(8)
    b.write_sam(gsam, header=False, threads=8)
    hits = b.hits()
    assert (hits["sa"] > np.uint64(2 ** 32)).any()          # rows above 2^32 were really visited
    g_l, o_l = _strip(gsam), _strip(osam)
    assert len(g_l) == n_reads
    bad = [i for i, (x, y) in enumerate(zip(g_l, o_l)) if x != y]
    assert not bad, (len(bad), g_l[bad[0]], o_l[bad[0]])
    assert (hits["type"] != 0).mean() > 0.9
    b.free()
