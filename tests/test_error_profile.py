"""Error-profile estimation between the two mapping passes (ErrorProfiling.java:100-631).
CPU tier: the C restatement of the Java loop (oracle/orc_profile.c) against hand-computed known answers.  GPU tier: `ps_error_profile` (HIP histogram over the records) writes byte-identical files from SAM
and from BAM input, and a --refine run uses the estimated profile.  Parity is unpinned: no fixture of the reference pairs an
input with an expected profile, and there is no JVM here to produce one."""
import os

import numpy as np
import pytest

import orc

FA = ">c1 some text\nACGTACGTACGTACGTACGTNNNNACGTACGTAAAACCCCGGGGTTTT\nACGTTGCA\n>c2\nTTTTTTTTTTGGGGGGGGGG\n"


def _sam(lines):
    return "@SQ\tSN:c1\tLN:56\n@SQ\tSN:c2\tLN:20\n" + "".join("\t".join(map(str, l)) + "\n" for l in lines)


def _run(tmp_path, lines, max_len=12):
    fa, sam = str(tmp_path / "r.fa"), str(tmp_path / "m.sam")
    open(fa, "w").write(FA)
    open(sam, "w").write(_sam(lines))
    n = orc.error_profile(sam, fa, max_len, sam)
    ep = [l.rstrip("\n").split("\t") for l in open(sam + ".errorprofile")]
    ip = open(sam + ".indelprofile").read()
    return n, ep, ip


def test_known_answer_plain_and_reverse(tmp_path):
    # forward read identical to c1[1..8] = ACGTACGT; reverse-strand record over c2[1..8] = TTTTTTTT whose SEQ (forward strand)
    # is TTTTCTTT: in read orientation the reference is AAAAAAAA and the read AAAGAAAA -> one A->G at read position 3
    n, ep, ip = _run(tmp_path, [("r1", 0, "c1", 1, 37, "8M", "*", 0, 0, "ACGTACGT", "IIIIIIII"),
                                ("r2", 16, "c2", 1, 37, "8M", "*", 0, 0, "TTTTCTTT", "IIIIIIII"),
                                ("r3", 4, "*", 0, 0, "*", "*", 0, 0, "ACGT", "IIII"),
                                ("r4", 1024, "c1", 1, 37, "8M", "*", 0, 0, "TTTTTTTT", "IIIIIIII"),
                                ("r5", 0, "c1", 0, 37, "8M", "*", 0, 0, "TTTTTTTT", "IIIIIIII")])
    assert n == 2                                   # unmapped, duplicate and position-less records are skipped (:155-166)
    assert all(len(r) == 5 and r[4] == "" for r in ep) and len(ep) == 4      # four values, each followed by a tab
    # A: 2 (r1) + 8 (r2) reference A's, one read as G; C, G, T: two each, all read as themselves
    assert ep[0][:4] == [orc.java_double(9 / 10), "0.0", orc.java_double(1 / 10), "0.0"]
    assert ep[1][:4] == ["0.0", "1.0", "0.0", "0.0"] and ep[2][:4] == ["0.0", "0.0", "1.0", "0.0"] and ep[3][:4] == ["0.0", "0.0", "0.0", "1.0"]
    assert ip == "0.0\t0.0"


def test_known_answer_holes_and_missing_bases(tmp_path):
    # c1[17..28] = ACGTNNNNACGT: the four N's are never counted (calculateArrayPos = -1); an N in the read neither;
    # no reference T is ever seen as anything -> still counted as T->T; a base never seen at all would print NaN
    n, ep, ip = _run(tmp_path, [("r1", 0, "c1", 17, 37, "12M", "*", 0, 0, "ACGTACGTACNT", "I" * 12)])
    assert n == 1
    assert ep[0][:4] == ["1.0", "0.0", "0.0", "0.0"] and ep[3][:4] == ["0.0", "0.0", "0.0", "1.0"]
    assert ep[2][:4] == ["0.0", "0.0", "1.0", "0.0"]          # G: c1[19], c1[27] -> the second is under the read's N: one count
    n, ep, ip = _run(tmp_path, [("r1", 0, "c2", 1, 37, "8M", "*", 0, 0, "TTTTTTTT", "I" * 8)])
    assert ep[0][:4] == ["NaN"] * 4 and ep[3][:4] == ["0.0", "0.0", "0.0", "1.0"]      # 0/0 for the bases never seen (:511-514)


def test_known_answer_indels(tmp_path):
    # insertion: read ACGTGACGT against c1[1..8] = ACGTACGT with 4M1I4M.  The spans differ (9 vs 8), so the alignment is
    # rebuilt from the CIGAR (:194-293): 8 match columns counted, the inserted base not; the gap is booked at column
    # (columns so far, insertion included) + 1 = 6 (:262-264).  Deletion: read ACGTCGT against c1[1..8], 4M1D3M: 7 columns
    # counted, gap at column 6 as well.
    n, ep, ip = _run(tmp_path, [("r1", 0, "c1", 1, 37, "4M1I4M", "*", 0, 0, "ACGTGACGT", "I" * 9),
                                ("r2", 0, "c1", 1, 37, "4M1D3M", "*", 0, 0, "ACGTCGT", "I" * 7),
                                ("r3", 0, "c1", 1, 37, "8M", "*", 0, 0, "ACGTACGT", "I" * 8)])
    assert n == 3
    assert [r[:4] for r in ep] == [["1.0", "0.0", "0.0", "0.0"], ["0.0", "1.0", "0.0", "0.0"], ["0.0", "0.0", "1.0", "0.0"], ["0.0", "0.0", "0.0", "1.0"]]
    # position 6 (read orientation, forward reads) was counted by r1 (column 6 = the A behind the insertion), r2 and r3: 3 bases;
    # one insertion and one deletion booked there: rate 1/3 each; every other position has rate 0 and is left out of the mean
    third = orc.java_double(1.0 / 3.0)
    assert ip == third + "\t" + third


def test_java_double_to_string():
    for v, s in ((0.99, "0.99"), (1e-4, "1.0E-4"), (2.1e-5, "2.1E-5"), (5.9e-4, "5.9E-4"), (0.001, "0.001"), (1.0, "1.0"), (0.0, "0.0"),
                 (1e7, "1.0E7"), (9999999.0, "9999999.0"), (0.12, "0.12"), (1.0 / 3.0, "0.3333333333333333"), (float("nan"), "NaN"), (123456.5, "123456.5")):
        assert orc.java_double(v) == s, v


def test_reference_example_profile_is_readable():
    """the reference's own example profile (hand-written input of its simulator: "0.990" style values, no trailing tab), when the
    tree is there -- it never travels to the GPU box: the reader the mapping step uses takes it as 16 numbers"""
    import ctypes as C
    p = "/root/reference/examples/simulation/example.errorprofile"
    if not os.path.exists(p):
        pytest.skip("reference tree not present")
    P = (C.c_double * 16)()
    a, b = C.c_double(), C.c_double()
    assert orc.lib().orc_read_profile_files(p.encode(), None, P, C.byref(a), C.byref(b)) == 0
    assert abs(sum(P[0:4]) - 1.0) < 1e-9 and abs(P[0] - 0.99) < 1e-12 and abs(P[15] - 0.987) < 1e-12


@pytest.mark.gpu
def test_hip_profile_identical_to_oracle(mid, workdir):
    """simulated PAR-CLIP reads with indels on both strands: first pass on the GPU, then the profile from its SAM and from the BAM
    made of it -- byte-identical to the restated Java loop; the refine pass then runs on the ESTIMATED profile."""
    import __graft_entry__ as ge
    import capi
    import simulate as S
    mod = ge.load_package()
    from conftest import sam_records
    sim = S.simulate_reads(mid["genome"], n_reads=30000, read_len=50, seed=91, indel_scale=400, n_frac=0.002)
    fq = os.path.join(workdir, "ep.fq")
    S.write_fastq(fq, sim)
    fa = mid["fa"]
    first = os.path.join(workdir, "ep_first")
    m1 = mod.mapping.BWAMapping()
    m1.executeMapping(8, fa, fq, first, 10, "2")
    sam = first + ".sam"
    recs = sam_records(sam)
    cig = [r.split("\t")[5] for r in recs]
    flg = [int(r.split("\t")[1]) for r in recs]
    assert sum("I" in c for c in cig) > 20 and sum("D" in c for c in cig) > 20                 # gapped hits ...
    assert sum(1 for c, f in zip(cig, flg) if ("I" in c or "D" in c) and f & 16) > 10            # ... on both strands
    assert sum(1 for f in flg if f & 4) > 10
    orc.error_profile(sam, fa, 101, os.path.join(workdir, "ep_orc"))
    exp_e, exp_i = open(os.path.join(workdir, "ep_orc.errorprofile"), "rb").read(), open(os.path.join(workdir, "ep_orc.indelprofile"), "rb").read()
    prof = mod.mapping.ErrorProfiling(sam, fa, 101)
    ep, ip = prof.inferErrorProfile(False, False)
    assert open(ep, "rb").read() == exp_e and open(ip, "rb").read() == exp_i
    vals = [[float(v) for v in l.split("\t")[:4]] for l in exp_e.decode().split("\n")[:4]]
    assert vals[3][1] > 0.01 and vals[3][1] > 3 * vals[0][1] and all(vals[j][j] > 0.8 for j in range(4))                       # the T->C conversions show; the diagonal dominates
    ins, dele = [float(v) for v in exp_i.decode().split("\t")]
    assert ins > 0 and dele > 0
    # the same from BAM (sorted, as the Java insists): counts do not depend on the order
    capi.ps_sam_to_bam(sam, first + ".bam", 0, True, True, 8)
    capi.ps_error_profile(first + ".bam", fa, 101, os.path.join(workdir, "ep_bam"))
    assert open(os.path.join(workdir, "ep_bam.errorprofile"), "rb").read() == exp_e
    assert open(os.path.join(workdir, "ep_bam.indelprofile"), "rb").read() == exp_i
    # the fused first pass: same SAM, and the profile of the MAPQ-filtered records counted from memory == the profile of the
    # filtered BAM file (what Main.java:327-334 would hand to ErrorProfiling), also for the oracle reading that file
    fused = os.path.join(workdir, "ep_fused")
    fe, fi = mod.mapping.BWAMapping().executeMappingWithProfile(8, fa, fq, fused, 30, "2", 101)
    assert sam_records(fused + ".sam") == recs
    capi.ps_sam_to_bam(sam, first + ".q30.bam", 30, True, True, 8)
    capi.ps_error_profile(first + ".q30.bam", fa, 101, os.path.join(workdir, "ep_q30"))
    q_e, q_i = open(os.path.join(workdir, "ep_q30.errorprofile"), "rb").read(), open(os.path.join(workdir, "ep_q30.indelprofile"), "rb").read()
    assert q_e != exp_e                                                  # the filter matters (MAPQ 25 = difference budget used up: those reads carry more errors)
    assert open(fe, "rb").read() == q_e and open(fi, "rb").read() == q_i
    mod.mapping.Mapping.filter_sam_mapq(sam, first + ".q30.sam", 30)
    orc.error_profile(first + ".q30.sam", fa, 101, os.path.join(workdir, "ep_q30_orc"))
    assert open(os.path.join(workdir, "ep_q30_orc.errorprofile"), "rb").read() == q_e
    assert open(os.path.join(workdir, "ep_q30_orc.indelprofile"), "rb").read() == q_i
    # a read longer than maxReadLength is an error (the Java's arrays would overflow), not a silent truncation
    with pytest.raises(capi.PsError):
        capi.ps_error_profile(sam, fa, 40, os.path.join(workdir, "ep_short"))
    # refine pass on the estimated profile == the oracle on the same files
    m2 = mod.mapping.PARAsuiteMapping()
    m2.setErrorProfileFilename(ep)
    m2.setIndelProfileFilename(ip)
    m2.executeMapping(8, fa, fq, os.path.join(workdir, "ep_refine"), 10, "-1")
    import ctypes as C
    P = (C.c_double * 16)()
    a, b = C.c_double(), C.c_double()
    assert orc.lib().orc_read_profile_files(ep.encode(), ip.encode(), P, C.byref(a), C.byref(b)) == 0
    osam = os.path.join(workdir, "ep_refine.orc.sam")
    mid["orc_index"].map_fastq(orc.profile_opt(list(P), a.value, b.value, -1), fq, osam, n_threads=8)
    assert sam_records(os.path.join(workdir, "ep_refine.sam")) == sam_records(osam)
