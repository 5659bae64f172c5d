"""GPU parity on a REPEAT-RICH genome: the stack tiers and the stop rules that an i.i.d. text never reaches.

bench.add_repeats lays the repeat structure of a mammalian genome over 128 Mbp of synthetic text: ~45 % of the bases inside
interspersed families (300-bp and 6-kb units, 2-20 % diverged copies), microsatellites, satellite arrays.  Reads cut from it
(and a batch cut only from inside the satellite arrays) run into what hg19 does to `bwa aln`:

  * searches whose stack outgrows the lane's private slice (large slot inside the launch), then the second narrow tier, then the
    wide tier with upstream's 2,000,000-entry limit -- asserted on the launch counters.  At 128 Mbp the deepest stack holds 60,000
    entries (33,750 at 32 Mbp: it grows with the copy numbers), just inside the 65,535 of a large slot and of the second tier: that
    tier is set to 32,768 entries and 32 hits per read here (default 65,535 / 256) and the launch gets 64 large slots instead of
    4,096 (PS_N_BIG), so that the wide tier runs on real repeat reads too; the first tier keeps its default size;
  * the `-R 30` rule (a worse hit arrives while more than 30 best ones are known: the search ends there) -- counted by the oracle
    on the same reads, and since hit lists and SAM are identical the product took it on the same reads;
  * hit lists of hundreds of SA intervals, X0 in the hundreds, XA lists cut at `-n 3`.

SAM and per-read hit lists must equal the oracle's (CPU restatement, parity unpinned; the oracle adopts the product's BWT at
this size -- its own builder is checked against the product's on the small genomes of test_gpu_parity.py)."""
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
def rep(tmp_path_factory):
    import torch
    import bench
    import capi
    import orc
    tmp = tmp_path_factory.mktemp("rep")
    dev = torch.device("cuda", 0)
    mbp = 128
    rinfo = {}
    contigs = bench.add_repeats(torch, dev, bench.gen_genome(torch, dev, mbp * 1_000_000, 4, 0x5EED0202), 0x5EED0209, info=rinfo, satellite_reps=2000)
    fa = str(tmp / "g.fa")
    bench.write_fasta(fa, contigs)
    os.environ["PS_N_BIG"] = "64"                         # read when a context is made (see the module docstring)
    try:
        ctx = capi.Ctx.build(fa, device=0)
    finally:
        del os.environ["PS_N_BIG"]
    info = ctx.info()
    oix = orc.Index.from_parts(fa, ctx.bwt_syms_chunked(), info.primary, ctx.sa_samples())
    return dict(torch=torch, dev=dev, contigs=contigs, fa=fa, ctx=ctx, oix=oix, tmp=tmp, satellites=rinfo["satellites"])


def _reads(rep, n, seed, inside_repeats):
    """50-bp simulated PAR-CLIP reads; inside_repeats: cut only from inside the satellite arrays (hundreds of 1-2 % diverged tandem
    copies of a 171-bp unit -- 2,000 here, the size of a small centromeric array: every such read has hundreds of near-identical places to go)"""
    import bench
    torch, dev, contigs = rep["torch"], rep["dev"], rep["contigs"]
    if not inside_repeats:
        return bench.gen_reads(torch, dev, contigs, n, 50, seed)
    sat = [("sat%d" % k, contigs[ci][1][at + 200:at + ln - 200].clone()) for k, (ci, at, ln) in enumerate(rep["satellites"])]
    sat = [(nm, c) for nm, c in sat if bool((c < 4).all())]  # an array that overlaps an N run of the contig was partly wiped: not used
    assert len(sat) >= 2
    return bench.gen_reads(torch, dev, sat, n, 50, seed)


@pytest.mark.parametrize("mode", ["stock", "profile"])
def test_repeat_rich_genome(rep, mode):
    import bench
    import orc
    ctx, oix, tmp = rep["ctx"], rep["oix"], rep["tmp"]
    P = np.array(bench.PROFILE)
    P[3, 1], P[3, 3] = 0.12, 0.87
    if mode == "stock":
        ctx.set_stock("0.04"); opt = orc.stock_opt("0.04")
    else:
        ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1); opt = orc.profile_opt(P, bench.INS_RATE, bench.DEL_RATE, -1)
    ctx.set_tiers(pool_cap=[16384, 32768, 2000064], aln_cap=[8, 32, 65536], bt_blocks=0)
    n = 60000
    codes = np.concatenate([_reads(rep, n - 6000, 0x5EED0203, False), _reads(rep, 6000, 0x5EED0204, True)])
    b = ctx.batch_from_codes(codes)
    b.run(8)
    tm = b.timing()
    gsam = str(tmp / ("rep_%s.gpu.sam" % mode))
    b.write_sam(gsam, header=False, threads=8)
    import simulate as S
    fq = str(tmp / ("rep_%s.fq" % mode))
    S.write_fastq(fq, dict(codes=codes, lens=np.full(n, 50, dtype=np.int32), quals=np.full((n, 50), 73, dtype=np.uint8)), names=["r%d" % i for i in range(n)])
    osam, osai = str(tmp / ("rep_%s.orc.sam" % mode)), str(tmp / ("rep_%s.orc.sai" % mode))
    orc.stats(reset=True)
    oix.map_fastq(opt, fq, osam, sai_out=osai, n_threads=16)
    st = orc.stats()
    print("repeat-rich %s: tier-1 overflow %d, tier-2 overflow %d reads; oracle: max stack %d, -R breaks %d, -m stops %d; backtrack %.0f ms in %d launches" %
          (mode, tm["n_overflow_tier1"], tm["n_overflow_tier2"], st["max_stack"], st["top2_breaks"], st["max_entries_stops"], tm["ms_backtrack"], tm["n_backtrack_launches"]))
    g, o = _strip(gsam), _strip(osam)
    assert len(g) == len(o) == n
    bad = [i for i in range(n) if g[i] != o[i]]
    assert not bad, (mode, len(bad), g[bad[0]], o[bad[0]])
    sai = orc.read_sai(osai)
    assert b.n_aln().tolist() == [len(x) for x in sai]
    # the shape was really exercised
    assert st["top2_breaks"] > 100, st
    assert st["max_stack"] > 32768, st                     # deeper than both narrow tiers as sized here
    assert tm["n_overflow_tier1"] > 0 and tm["n_overflow_tier2"] > 0, tm
    ctx.set_tiers(pool_cap=[16384, 65535, 2000064], aln_cap=[8, 256, 65536], bt_blocks=0)
    hits = b.hits()
    assert (hits["c1"] > 30).mean() > 0.01 and (hits["c1"] > 100).sum() > 100
