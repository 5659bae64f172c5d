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


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    """the 2.2 Gbp genome, its index (checked against the text, see below), the text and the oracle's adopted index: built once"""
    import torch
    import bench
    import capi
    import orc
    sys.path.insert(0, HERE)
    import index_props as IP
    tmp = tmp_path_factory.mktemp("big")
    dev = torch.device("cuda", 0)
    mbp = 2200
    contigs = bench.gen_genome(torch, dev, mbp * 1_000_000, 6, 0x5EED0102)
    fa = str(tmp / "g.fa")
    bench.write_fasta(fa, contigs)
    torch.cuda.empty_cache()
    ctx = capi.Ctx.build(fa, device=0)
    info = ctx.info()
    assert info.seq_len == 2 * mbp * 1_000_000 and info.seq_len > 2 ** 32
    assert sum(int(info.L2[c + 1]) - int(info.L2[c]) for c in range(4)) == info.seq_len
    # the text: the packed forward strand, itself compared with the FASTA codes at every non-N position
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
    bwt_all = ctx.bwt_syms_chunked()
    oix = orc.Index.from_parts(fa, bwt_all, info.primary, ctx.sa_samples())     # adopted -- test_index_from_the_text checks it
    return dict(torch=torch, dev=dev, contigs=contigs, fa=fa, ctx=ctx, info=info, text=IP.Text(fwd), bwt_all=bwt_all, oix=oix, tmp=tmp)


def test_every_row_against_the_text(big):
    """all 4.4e9 rows, not a sample: the LF cycle of the index spells the packed text (compared with the FASTA in the fixture) and
    every SA sample holds the position counted along it (ps_ctx_index_check; tests/test_gpu_index_check.py has the negative controls)"""
    r = big["ctx"].index_check()
    assert r["rows"] == big["info"].seq_len + 1, r
    assert r["bad_symbols"] == 0 and r["bad_samples"] == 0, r
    print("index check at 2.2 Gbp: %d rows visited, longest arc between two SA samples %d rows" % (r["rows"], r["longest_arc"]))


def test_index_from_the_text(big):
    """the index checked WITHOUT trusting it: properties computed from the text (tests/index_props.py; the same check fails on a
    suffix array with unsorted ties, swapped neighbours or a wrong position bit: tests/test_index_props_cpu.py)"""
    import index_props as IP
    ctx, info, bwt_all = big["ctx"], big["info"], big["bwt_all"]
    rng = np.random.default_rng(5)
    rows = np.concatenate([rng.integers(1, info.seq_len - 1, 120_000), np.arange(2 ** 32 - 1500, 2 ** 32 + 1500)]).astype(np.int64)
    primary = int(info.primary)

    def bwt_lookup(r):
        r = np.asarray(r, dtype=np.int64)
        out = bwt_all[np.minimum(r - (r >= primary), bwt_all.size - 1)].astype(np.int16)
        out[r == primary] = 255
        return out
    seen = IP.check_index(big["text"], rows, lambda r: ctx.sa_lookup(r), bwt_lookup, primary, [int(info.L2[c]) for c in range(5)])
    assert seen["pairs"] == rows.size and seen["above_2_32"] > 1000
    assert seen["long_lcp"] > 1000 and seen["max_lcp"] > 500       # pairs inside the 2 kb two-copy segments: what only the late doubling rounds order


def test_exact_reads_1m_map_home(big):
    """BASELINE configs[1] at a meaningful size: 1 M x 50 bp exact reads (stock -n 0: the seed-only path) against 4.4e9 rows.
    Size-independent property: every read is placed, and the reference at the reported place spells the read (its origin, or an
    equal copy in a two-copy segment), on the reported strand.  Plus equality with the oracle on a 50,000-read sample."""
    import orc
    import simulate as S
    torch, dev, ctx, contigs = big["torch"], big["dev"], big["ctx"], big["contigs"]
    n, L = 1_000_000, 50
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    flat = torch.cat([c for _, c in contigs])
    starts = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * (flat.numel() - L - 1)).long()
    ar = torch.arange(L, device=dev)[None, :]
    for _ in range(6):                                         # re-draw windows that touch an N run (or a contig end: harmless, exact anyway?)
        win = flat[starts[:, None] + ar]
        bad = (win == 4).any(1)
        if not bool(bad.any()):
            break
        starts[bad] = (torch.rand(int(bad.sum()), generator=g, device=dev, dtype=torch.float64) * (flat.numel() - L - 1)).long()
    win = flat[starts[:, None] + ar]
    keep = (win < 4).all(1)
    # windows must lie inside one contig
    bounds = torch.cumsum(torch.tensor([c.numel() for _, c in contigs], device=dev), 0)
    cid_a = torch.searchsorted(bounds, starts, right=True); cid_b = torch.searchsorted(bounds, starts + L - 1, right=True)
    keep &= cid_a == cid_b
    win = win[keep]
    rev = torch.rand(win.shape[0], generator=g, device=dev) < 0.5
    reads = torch.where(rev[:, None], 3 - win.flip(1), win).cpu().numpy()
    del flat
    ctx.set_stock("0")
    b = ctx.batch_from_codes(np.ascontiguousarray(reads))
    b.run(8)
    hits = b.hits()
    assert (hits["type"] != 0).all()                            # every exact read is placed
    text = big["text"]
    pos = hits["pos"].astype(np.int64)
    ref = text.fwd[pos[:, None] + np.arange(L)[None, :]]
    spelled = np.where((hits["strand"] != 0)[:, None], 3 - ref[:, ::-1], ref)
    assert np.array_equal(spelled, reads)                       # ... where the reference spells it, on the strand reported
    assert (hits["n_mm"] == 0).all() and (hits["n_gapo"] == 0).all()
    multi = (hits["c1"] > 1).sum()
    assert multi > 1000                                         # reads from the two-copy segments: either copy is a right answer
    b.free()
    ns = 50_000
    sim = dict(codes=reads[:ns], lens=np.full(ns, L, dtype=np.int32), quals=np.full((ns, L), 73, dtype=np.uint8))
    fq = str(big["tmp"] / "exact.fq")
    S.write_fastq(fq, sim, names=["r%d" % i for i in range(ns)])
    osam, gsam = str(big["tmp"] / "exact.o.sam"), str(big["tmp"] / "exact.g.sam")
    big["oix"].map_fastq(orc.stock_opt("0"), fq, osam, n_threads=16)
    sb = ctx.batch_from_codes(np.ascontiguousarray(reads[:ns]))
    sb.run(8)
    sb.write_sam(gsam, header=False, threads=8)
    assert _strip(gsam) == _strip(osam)
    sb.free()


def test_ragged_36_75_with_indel_profile_200k(big):
    """BASELINE configs[4]'s shape at a meaningful size on one GPU: 200,000 reads of 36-75 bp (ragged: several cost classes, each a
    launch with per-lane lengths), error-profile costs with the indel profile, gapped extension -- SAM equal to the oracle's."""
    import bench
    import orc
    import simulate as S
    torch, dev, ctx, contigs = big["torch"], big["dev"], big["ctx"], big["contigs"]
    n, Lmax = 200_000, 75
    P = np.array(bench.PROFILE)
    P[3, 1], P[3, 3] = 0.12, 0.87
    ins, dele = 10 * bench.INS_RATE, 10 * bench.DEL_RATE       # a profile with visibly cheaper gaps than the default
    ctx.set_profile(P, ins, dele, -1)
    rd = bench.gen_reads(torch, dev, contigs, n, Lmax, 0x5EED0105, indels=True)
    lens = np.random.default_rng(3).integers(36, Lmax + 1, n).astype(np.int32)
    fq = str(big["tmp"] / "ragged.fq")
    S.write_fastq(fq, dict(codes=rd, lens=lens, quals=np.full((n, Lmax), 73, dtype=np.uint8)), names=["r%d" % i for i in range(n)])
    osam, gsam = str(big["tmp"] / "ragged.o.sam"), str(big["tmp"] / "ragged.g.sam")
    big["oix"].map_fastq(orc.profile_opt(P, ins, dele, -1), fq, osam, n_threads=16)
    b = ctx.batch_from_fastq(fq)
    b.run(8)
    b.write_sam(gsam, header=False, threads=8)
    g_l, o_l = [l for l in open(gsam) if not l.startswith("@")], [l for l in open(osam) if not l.startswith("@")]
    assert len(g_l) == n
    bad = [i for i, (x, y) in enumerate(zip(g_l, o_l)) if x != y]
    assert not bad, (len(bad), g_l[bad[0]], o_l[bad[0]])
    hits = b.hits()
    assert (hits["type"] != 0).mean() > 0.85 and (hits["n_gapo"] > 0).sum() > 50
    assert b.timing()["n_backtrack_launches"] == 1            # 40 lengths, three difference budgets, three packed word counts: ONE launch
                                                               # (every read carries its own budget; it used to be one launch per cost class)
    b.free()


def test_parity_and_round_trip_above_2_pow_32_rows(big, tmp_path):
    import bench
    import orc
    import simulate as S
    torch, dev, ctx, contigs, info = big["torch"], big["dev"], big["ctx"], big["contigs"], big["info"]
    n_reads, L = 30000, 50
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
    sim = dict(codes=rd, lens=np.full(n_reads, L, dtype=np.int32), quals=np.full((n_reads, L), 73, dtype=np.uint8))
    fq = str(tmp_path / "r.fq")
    S.write_fastq(fq, sim, names=["r%d" % i for i in range(n_reads)])
    oix = big["oix"]
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
