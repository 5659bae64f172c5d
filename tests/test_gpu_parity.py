"""GPU parity: libparasuite_hip.so (through its C ABI) against the CPU oracle, bit for bit.

The oracle's status is PARITY UNPINNED (oracle/ps_oracle.h): the reference tree holds no aligner
source, tests or golden SAM, so "identical to the reference" here means identical to this
repository's restatement of BWA-0.7.8-style aln+samse semantics.
"""
import os

import numpy as np
import pytest

from conftest import sam_records, sam_sq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx_example(example):
    import capi
    return capi.Ctx.build(example["fa"])


@pytest.fixture(scope="module")
def ctx_multi(multi):
    import capi
    return capi.Ctx.build(multi["fa"])


def _check_index(ctx, oix):
    info = ctx.info()
    assert info.seq_len == oix.seq_len and info.l_pac == oix.l_pac
    assert info.primary == oix.primary
    assert list(info.L2) == oix.L2
    syms, cnts = ctx.bwt_syms()
    ref = oix.bwt_syms()
    assert np.array_equal(syms, ref)
    # block counts = running symbol counts at every 192-symbol boundary
    run = np.zeros(4, dtype=np.int64)
    onehot = np.stack([(ref == c) for c in range(4)], axis=1).astype(np.int64)
    pre = np.concatenate([np.zeros((1, 4), dtype=np.int64), np.cumsum(onehot, axis=0)])
    idx = np.minimum(np.arange(cnts.shape[0], dtype=np.int64) * 192, ref.size)
    assert np.array_equal(cnts.astype(np.int64), pre[idx])
    sa = ctx.sa_samples()
    osa = oix.sa_samples()
    assert sa.size == osa.size
    assert np.array_equal(sa[1:], osa[1:])          # entry 0 (the empty suffix) is never asked for
    assert np.array_equal(ctx.fetch(2), oix.pac())


def test_index_matches_oracle(ctx_example, example):
    _check_index(ctx_example, example["orc_index"])


def test_index_multi_contig(ctx_multi, multi):
    _check_index(ctx_multi, multi["orc_index"])
    assert ctx_multi.info().n_contigs == 3


def test_fasta_parsed_in_pieces(multi, monkeypatch):
    """the FASTA parser works on pieces of whole lines in parallel (3 GB references); with pieces of ~100 bytes every N run
    and IUPAC hole of the fixture crosses piece boundaries, and the lrand48 fill must continue mid-stream"""
    import capi
    monkeypatch.setenv("PS_FASTA_PIECE", "100")
    ctx = capi.Ctx.build(multi["fa"], device=0)
    _check_index(ctx, multi["orc_index"])
    ometa = multi["orc_index"]
    meta = ctx.meta().decode()
    monkeypatch.delenv("PS_FASTA_PIECE")
    ref = capi.Ctx.build(multi["fa"], device=0)
    assert meta == ref.meta().decode()                      # contig table and hole list identical to the one-piece parse
    ctx.close(); ref.close()


def _aln_tuple(a):
    return (int(a["k"]), int(a["l"]), int(a["n_mm"]), int(a["n_gapo"]), int(a["n_gape"]), int(a["n_ins"]),
            int(a["n_del"]), int(a["score"]))


def _compare(ctx, oix, opt, fq, workdir, tag, n_check_alns=400):
    import orc
    b = ctx.batch_from_fastq(fq)
    b.run(threads=4)
    gsam = os.path.join(workdir, tag + ".gpu.sam")
    osam = os.path.join(workdir, tag + ".orc.sam")
    osai = os.path.join(workdir, tag + ".orc.sai")
    b.write_sam(gsam)
    oix.map_fastq(opt, fq, osam, sai_out=osai, n_threads=8)
    sai = orc.read_sai(osai)
    n_aln = b.n_aln()
    assert n_aln.tolist() == [len(x) for x in sai], tag
    for r in range(min(len(sai), n_check_alns)):
        got = [_aln_tuple(a) for a in b.alns(r)]
        exp = [(int(a["k"]), int(a["l"]), int(a["n_mm"]), int(a["n_gapo"]), int(a["n_gape"]), int(a["n_ins"]),
                int(a["n_del"]), int(a["score"])) for a in sai[r]]
        assert got == exp, (tag, r)
    assert sam_sq(gsam) == sam_sq(osam)
    g, o = sam_records(gsam), sam_records(osam)
    assert len(g) == len(o)
    bad = [i for i in range(len(g)) if g[i] != o[i]]
    assert not bad, (tag, len(bad), g[bad[0]], o[bad[0]])
    return b


def _fastq(genome, workdir, name, **kw):
    import simulate as S
    sim = S.simulate_reads(genome, **kw)
    fq = os.path.join(workdir, name + ".fq")
    S.write_fastq(fq, sim)
    return fq


@pytest.mark.parametrize("n_arg", ["0.04", "2", "0"])
def test_stock_sam_bit_exact(ctx_example, example, workdir, n_arg):
    import orc
    fq = _fastq(example["genome"], workdir, "stock50", n_reads=3000, read_len=50, seed=11, indel_scale=30, n_frac=0.002)
    ctx_example.set_stock(n_arg)
    _compare(ctx_example, example["orc_index"], orc.stock_opt(n_arg), fq, workdir, "stock_" + n_arg)


@pytest.mark.parametrize("x", [-1, 2])
def test_profile_sam_bit_exact(ctx_example, example, workdir, x):
    import orc
    import simulate as S
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87          # a PAR-CLIP like T->C rate
    fq = _fastq(example["genome"], workdir, "prof50", n_reads=3000, read_len=50, seed=12, indel_scale=30)
    ctx_example.set_profile(P, 2.1e-5, 5.9e-4, x)
    _compare(ctx_example, example["orc_index"], orc.profile_opt(P, 2.1e-5, 5.9e-4, x), fq, workdir, "prof_%d" % x)


@pytest.mark.parametrize("mode", ["stock", "profile"])
def test_mixed_lengths_and_contig_edges(ctx_multi, multi, workdir, mode):
    """ragged input (36-75 bp, length-binned on the device: BASELINE configs[4] in small), multi contig, repeats, N runs;
    stock costs and the error + indel profile"""
    import orc
    import simulate as S
    fq = _fastq(multi["genome"], workdir, "mixed", n_reads=2500, read_len=75, min_len=36, seed=13, indel_scale=40,
                n_frac=0.003)
    if mode == "stock":
        ctx_multi.set_stock("0.04")
        opt = orc.stock_opt("0.04")
    else:
        P = S.EXAMPLE_PROFILE.copy()
        P[3, 1], P[3, 3] = 0.12, 0.87
        ctx_multi.set_profile(P, 2.1e-5, 5.9e-4, -1)
        opt = orc.profile_opt(P, 2.1e-5, 5.9e-4, -1)
    _compare(ctx_multi, multi["orc_index"], opt, fq, workdir, "mixed_" + mode)


def test_small_tiers_escalate(ctx_example, example, workdir):
    """tiny stack/hit capacities force the larger search tiers; results must not change"""
    import orc
    fq = _fastq(example["genome"], workdir, "tiers", n_reads=600, read_len=50, seed=14, indel_scale=30)
    ctx_example.set_stock("0.04")
    ctx_example.set_tiers(pool_cap=[48, 4096, 2000064], aln_cap=[1, 64, 65536], bt_blocks=2)
    try:
        b = _compare(ctx_example, example["orc_index"], orc.stock_opt("0.04"), fq, workdir, "tiers")
        assert b.timing()["n_overflow_tier1"] > 0
    finally:
        ctx_example.set_tiers(pool_cap=[16384, 65535, 2000064], aln_cap=[8, 256, 65536], bt_blocks=0)


def test_empty_and_degenerate_reads(ctx_example, example, workdir):
    import orc
    fq = os.path.join(workdir, "degenerate.fq")
    with open(fq, "w") as f:
        f.write("@allN\n" + "N" * 40 + "\n+\n" + "I" * 40 + "\n")
        f.write("@polyA/1\n" + "A" * 50 + "\n+\n" + "I" * 50 + "\n")
        f.write("@short\nACGTACGTACGTAC\n+\nIIIIIIIIIIIIII\n")
        f.write("@one\nA\n+\nI\n")
    ctx_example.set_stock("0.04")
    _compare(ctx_example, example["orc_index"], orc.stock_opt("0.04"), fq, workdir, "degenerate")


@pytest.fixture(scope="module")
def ctx_mid(mid):
    import capi
    return capi.Ctx.build(mid["fa"])


def test_index_mid_genome(ctx_mid, mid):
    _check_index(ctx_mid, mid["orc_index"])


@pytest.mark.parametrize("read_len,mode", [(36, "stock"), (75, "stock"), (100, "stock"), (36, "profile"), (75, "profile"), (100, "profile")])
def test_lengths_and_modes_mid_genome(ctx_mid, mid, workdir, read_len, mode):
    """the read lengths of BASELINE.json configs[4] (36-75 bp) and 100 bp, both cost models, 8 Mbp genome"""
    import orc
    import simulate as S
    fq = _fastq(mid["genome"], workdir, "mid_%d" % read_len, n_reads=4000, read_len=read_len, seed=100 + read_len, indel_scale=4.0)
    if mode == "stock":
        ctx_mid.set_stock("0.04")
        opt = orc.stock_opt("0.04")
    else:
        P = S.EXAMPLE_PROFILE.copy()
        P[3, 1], P[3, 3] = 0.12, 0.87
        ctx_mid.set_profile(P, 2.1e-5, 5.9e-4, -1)
        opt = orc.profile_opt(P, 2.1e-5, 5.9e-4, -1)
    b = _compare(ctx_mid, mid["orc_index"], opt, fq, workdir, "mid_%d_%s" % (read_len, mode), n_check_alns=200)
    h = b.hits()
    assert (h["type"] != 0).mean() > 0.8


def test_reference_accuracy_rule(ctx_mid, mid, workdir):
    """the reference's own acceptance rule for simulated reads (ValidateBenchmarkStatisticsPARCLIP.java:145-159:
    same chromosome, start-5 <= alignmentStart, end+5 >= alignmentEnd) on the product's SAM"""
    import simulate as S
    sim = S.simulate_reads(mid["genome"], 6000, 50, seed=321, indel_scale=1.0)
    fq = os.path.join(workdir, "acc.fq")
    S.write_fastq(fq, sim)
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    ctx_mid.set_profile(P, 2.1e-5, 5.9e-4, -1)
    b = ctx_mid.batch_from_fastq(fq)
    b.run(4)
    sam = os.path.join(workdir, "acc.sam")
    b.write_sam(sam)
    mapped, correct, total = S.score_truth(sam)
    assert total == 6000 and mapped > 0.9 * total and correct > 0.99 * mapped


def test_config0_refine_run_100k_reads(example, workdir):
    """BASELINE configs[0]'s shape -- ~100 k x 50 bp simulated PAR-CLIP reads against the (regenerated) example reference -- as the
    whole `--refine` sequence of Main.java:283-340 through the mirror classes: first pass with stock costs, error profile from its
    MAPQ-filtered alignments (fused into the pass), second pass with that profile.  Every file is compared with the oracle run on
    the same inputs: both SAM files line by line, the two profile files byte for byte."""
    import ctypes as C
    import shutil
    import __graft_entry__ as ge
    import orc
    import simulate as S
    mod = ge.load_package()
    d = os.path.join(workdir, "cfg0")
    os.makedirs(d, exist_ok=True)
    fa = os.path.join(d, "example.fa")
    shutil.copy(example["fa"], fa)
    sim = S.simulate_reads(example["genome"], n_reads=100000, read_len=50, seed=1000, indel_scale=100, n_frac=0.001)
    fq = os.path.join(d, "reads.fq")
    S.write_fastq(fq, sim)
    oix = example["orc_index"]
    # first pass (BWAMapping: stock `aln -n 2`) + the profile of its MAPQ >= 10 alignments
    m1 = mod.mapping.BWAMapping()
    ep, ip = m1.executeMappingWithProfile(8, fa, fq, os.path.join(d, "first"), 10, "2", 101)
    o1 = os.path.join(d, "first.orc.sam")
    oix.map_fastq(orc.stock_opt("2"), fq, o1, n_threads=8)
    assert sam_records(os.path.join(d, "first.sam")) == sam_records(o1)
    mod.mapping.Mapping.filter_sam_mapq(o1, os.path.join(d, "first.orc.q10.sam"), 10)
    orc.error_profile(os.path.join(d, "first.orc.q10.sam"), fa, 101, os.path.join(d, "orc"))
    assert open(ep, "rb").read() == open(os.path.join(d, "orc.errorprofile"), "rb").read()
    assert open(ip, "rb").read() == open(os.path.join(d, "orc.indelprofile"), "rb").read()
    # refine pass (PARAsuiteMapping) on the estimated profile
    m2 = mod.mapping.PARAsuiteMapping()
    m2.setErrorProfileFilename(ep)
    m2.setIndelProfileFilename(ip)
    m2.executeMapping(8, fa, fq, os.path.join(d, "refine"), 10, "-1")
    P = (C.c_double * 16)()
    a, b = C.c_double(), C.c_double()
    assert orc.lib().orc_read_profile_files(ep.encode(), ip.encode(), P, C.byref(a), C.byref(b)) == 0
    o2 = os.path.join(d, "refine.orc.sam")
    oix.map_fastq(orc.profile_opt(list(P), a.value, b.value, -1), fq, o2, n_threads=8)
    got = sam_records(os.path.join(d, "refine.sam"))
    assert got == sam_records(o2)
    mapped = sum(1 for l in got if not int(l.split("\t")[1]) & 4)
    assert mapped > 0.7 * len(got) and len(got) == 100000


def test_stack_growth_in_launch(ctx_example, example, workdir):
    """a private stack slice of 64 entries: almost every read moves to a large slot inside the launch
    (wave-cooperative copy) and none needs the next tier; results unchanged"""
    import orc
    fq = _fastq(example["genome"], workdir, "grow", n_reads=1500, read_len=50, seed=15, indel_scale=30)
    ctx_example.set_stock("0.04")
    ctx_example.set_tiers(pool_cap=[64, 65535, 2000064], aln_cap=[8, 256, 65536], bt_blocks=0)
    try:
        b = _compare(ctx_example, example["orc_index"], orc.stock_opt("0.04"), fq, workdir, "grow")
        assert b.timing()["n_overflow_tier1"] == 0
    finally:
        ctx_example.set_tiers(pool_cap=[16384, 65535, 2000064], aln_cap=[8, 256, 65536], bt_blocks=0)


def test_ps_map_streams_in_pieces(mid, workdir, monkeypatch, capfd):
    """ps_map streams the input in pieces that a parser thread, the GPU workers (two per device by default, each with its own
    stream and workspace) and a SAM writer work on side by side; the output must not depend on the cut or on the number of
    workers (the tie-break stream is carried from piece to piece in input order) and equals the oracle's."""
    import re
    import capi
    import orc
    import simulate as S
    sim = S.simulate_reads(mid["genome"], n_reads=40000, read_len=50, seed=77, indel_scale=30, n_frac=0.001)
    fq = os.path.join(workdir, "stream.fq")
    S.write_fastq(fq, sim)
    assert os.path.getsize(fq) > 4 << 20
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    ep, ip = os.path.join(workdir, "s.errorprofile"), os.path.join(workdir, "s.indelprofile")
    with open(ep, "w") as f:
        for row in P:
            f.write("".join(repr(float(v)) + "\t" for v in row) + "\n")
    open(ip, "w").write("2.1E-5\t5.9E-4")
    fa = mid["fa"]
    if not os.path.exists(fa + ".bwt"):
        capi.ps_index(fa)
    outs = []
    monkeypatch.setenv("PS_VERBOSE", "1")
    for tag, mb, ids, per in (("one", "4096", None, "1"), ("many", "1", None, "1"), ("two_workers", "1", None, "2"), ("named_twice", "1", "0,0", "1")):
        monkeypatch.setenv("PS_CHUNK_MB", mb)
        monkeypatch.setenv("PS_WORKERS_PER_GPU", per)
        if ids:                                   # a device named twice: two workers on it, one copy of the index
            monkeypatch.setenv("PARASUITE_GPU_IDS", ids)
        out = os.path.join(workdir, "stream_%s.sam" % tag)
        capfd.readouterr()
        capi.ps_map(8, "-1", ep, ip, fa, fq, out)
        err = capfd.readouterr().err
        outs.append(open(out, "rb").read())
        pieces = re.findall(r"piece (\d+) on device 0 worker (\d+)", err)
        if tag == "one":
            assert len(pieces) == 1
        else:
            assert len(pieces) >= 4
        if tag in ("two_workers", "named_twice"):   # pieces are searched out of order by two workers: chain and writer restore it
            assert {w for _, w in pieces} == {"0", "1"}, err
            assert "1 device(s) x 2 worker(s)" in err
    monkeypatch.delenv("PARASUITE_GPU_IDS")
    assert outs[0] == outs[1] == outs[2] == outs[3]
    osam = os.path.join(workdir, "stream.orc.sam")
    mid["orc_index"].map_fastq(orc.profile_opt(P, 2.1e-5, 5.9e-4, -1), fq, osam, n_threads=8)
    g, o = sam_records(os.path.join(workdir, "stream_many.sam")), sam_records(osam)
    assert len(g) == len(o) == 40000
    bad = [i for i in range(len(g)) if g[i] != o[i]]
    assert not bad, (len(bad), g[bad[0]], o[bad[0]])


def test_cloned_index_maps_identically(ctx_multi, multi, workdir):
    """every device after the first gets its index from the first one, device to device (index_clone: blobs + the 2.9 GB jump
    table at hg19 size).  One-GPU rehearsal: a second context on the SAME device with a cloned index must be the same index --
    every row against the text (ps_ctx_index_check), the same meta / jump levels -- and map to the same SAM.  (Between two real
    devices the copy is hipMemcpyPeerAsync over xGMI: that leg has no hardware run)"""
    import orc
    c2 = ctx_multi.clone(0)
    i1, i2 = ctx_multi.info(), c2.info()
    assert (i1.seq_len, i1.primary, list(i1.L2), i1.n_contigs, i1.jump_levels) == (i2.seq_len, i2.primary, list(i2.L2), i2.n_contigs, i2.jump_levels)
    assert i2.jump_levels > 0
    assert ctx_multi.meta() == c2.meta()
    for k in range(3):
        assert np.array_equal(ctx_multi.fetch(k), c2.fetch(k))
    r = c2.index_check()
    assert r["rows"] == i2.seq_len + 1 and r["bad_symbols"] == 0 and r["bad_samples"] == 0
    fq = _fastq(multi["genome"], workdir, "clone", n_reads=3000, read_len=50, seed=17, indel_scale=30)
    c2.set_stock("0.04")
    _compare(c2, multi["orc_index"], orc.stock_opt("0.04"), fq, workdir, "clone")
    c2.close()


def test_ps_map_empty_fastq_writes_header(multi, workdir):
    """an input without reads: upstream's samse prints the @SQ header before its read loop (oracle/ps_oracle.c: orc_map_fastq
    does the same), so `samtools view -bS` (PARAsuiteMapping.java:103-110) still gets a valid SAM: ps_map must not leave 0 bytes"""
    import capi
    import orc
    fa = multi["fa"]
    if not os.path.exists(fa + ".bwt"):
        capi.ps_index(fa)
    for tag, text in (("empty", ""), ("blank", "\n\n")):
        fq = os.path.join(workdir, "none_%s.fq" % tag)
        open(fq, "w").write(text)
        out, osam = os.path.join(workdir, "none_%s.sam" % tag), os.path.join(workdir, "none_%s.orc.sam" % tag)
        capi.ps_map(4, "0.04", None, None, fa, fq, out)
        multi["orc_index"].map_fastq(orc.stock_opt("0.04"), fq, osam, n_threads=2)
        assert sam_records(out) == sam_records(osam) == []
        assert sam_sq(out) == sam_sq(osam) and len(sam_sq(out)) == 3
        assert [l.split("\t")[1] for l in sam_sq(out)] == ["SN:chrA", "SN:chrB", "SN:chrC"]         # FASTA order
    # ... and through the argv shim, which is what the unmodified jar would run (PARAsuiteMapping.java:63-92)
    import subprocess
    bwa = os.path.join(os.path.dirname(capi.__file__), "bin", "bwa")
    fq = os.path.join(workdir, "none_empty.fq")
    sai, out = os.path.join(workdir, "none.sai"), os.path.join(workdir, "none_shim.sam")
    subprocess.check_call([bwa, "aln", "-t", "2", "-n", "2", fa, fq, "-f", sai])
    subprocess.check_call([bwa, "samse", fa, sai, fq, "-f", out])
    assert sam_records(out) == [] and len(sam_sq(out)) == 3


def test_ps_map_reports_errors_without_hanging(mid, workdir):
    """failures in the parser or the writer thread of ps_map come back as an error status (Mapping.executeCommand's
    contract: non-zero, message) -- and the other stages are released, no thread waits forever"""
    import capi
    fa = mid["fa"]
    if not os.path.exists(fa + ".bwt"):
        capi.ps_index(fa)
    fq = os.path.join(workdir, "stream.fq")                # written by the streaming test; recreate if run alone
    if not os.path.exists(fq):
        import simulate as S
        S.write_fastq(fq, S.simulate_reads(mid["genome"], n_reads=40000, read_len=50, seed=77))
    with pytest.raises(capi.PsError):
        capi.ps_map(4, "0.04", None, None, fa, os.path.join(workdir, "no_such.fq"), os.path.join(workdir, "x.sam"))
    os.environ["PS_CHUNK_MB"] = "1"
    try:
        with pytest.raises(capi.PsError):
            capi.ps_map(4, "0.04", None, None, fa, fq, os.path.join(workdir, "no_such_dir", "x.sam"))
    finally:
        del os.environ["PS_CHUNK_MB"]
    with pytest.raises(capi.PsError):
        capi.ps_map(4, "0.04", os.path.join(workdir, "no_such.errorprofile"), None, fa, fq, os.path.join(workdir, "y.sam"))


def test_estimated_best_score_spares_entries_and_never_changes_results(ctx_mid, mid, workdir, monkeypatch):
    """The search kernel leaves out children that can only matter if the read's best hit is worse than the score estimated for it
    (ps_narrow.h: nt_tail; first tier, profile costs) and starts a read over without the estimate when that fails.  Same hits and
    SAM as the oracle with the estimate in force, without it (PS_CAP=0) and with estimates made too low by 8 and by 24 units
    (PS_CAP_BIAS: the restart path runs for most reads); the counting kernel shows that entries were really spared / re-made."""
    import orc
    import simulate as S
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    fq = _fastq(mid["genome"], workdir, "cap50", n_reads=12000, read_len=50, seed=77, indel_scale=6.0, n_frac=0.002)
    ctx_mid.set_profile(P, 2.1e-5, 5.9e-4, -1)
    opt = orc.profile_opt(P, 2.1e-5, 5.9e-4, -1)
    pushes = {}
    ctx_mid.set_stats(True)
    try:
        for tag, env in (("cap", {}), ("nocap", {"PS_CAP": "0"}), ("low8", {"PS_CAP_BIAS": "8"}), ("low24", {"PS_CAP_BIAS": "24"})):
            for k in ("PS_CAP", "PS_CAP_BIAS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            b = _compare(ctx_mid, mid["orc_index"], opt, fq, workdir, "cap_" + tag, n_check_alns=300)
            assert b.timing()["n_backtrack_launches"] == 1, tag            # a failed estimate is handled inside the launch
            pushes[tag] = b.kstats(1)["pushes"]
    finally:
        ctx_mid.set_stats(False)
    print("entries stored:", pushes)
    assert pushes["cap"] < 0.9 * pushes["nocap"], pushes                 # the estimate spares entries ...
    assert pushes["low24"] > pushes["cap"], pushes                       # ... and a failed one costs a second search of the read


@pytest.mark.parametrize("mode", ["stock", "profile"])
def test_lengths_on_both_sides_of_the_seed_share_a_launch(ctx_mid, mid, workdir, mode):
    """adapter-trimmed PAR-CLIP reads of 18-44 bases: upstream gives a read no longer than the 32-base seed no seed rule; the
    kernels test that per read (len > seed_len), so that both kinds are ONE launch -- SAM and hit lists == oracle"""
    import orc
    import simulate as S
    fq = _fastq(mid["genome"], workdir, "seedmix", n_reads=6000, read_len=44, min_len=18, seed=314, indel_scale=6.0, n_frac=0.003)
    if mode == "stock":
        ctx_mid.set_stock("0.04")
        opt = orc.stock_opt("0.04")
    else:
        P = S.EXAMPLE_PROFILE.copy()
        P[3, 1], P[3, 3] = 0.12, 0.87
        ctx_mid.set_profile(P, 2.1e-5, 5.9e-4, -1)
        opt = orc.profile_opt(P, 2.1e-5, 5.9e-4, -1)
    b = _compare(ctx_mid, mid["orc_index"], opt, fq, workdir, "seedmix_" + mode, n_check_alns=400)
    tm = b.timing()
    if mode == "profile":                      # one gap limit for every length: one bin, one first-tier launch (+ one per larger tier that was needed)
        assert tm["n_backtrack_launches"] == 1 + (tm["n_overflow_tier1"] > 0) + (tm["n_overflow_tier2"] > 0), tm
