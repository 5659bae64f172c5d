"""Randomised parity sweep on the GPU: product (through the C ABI) against the CPU oracle, SAM line by line and hit lists.

Each case draws a genome shape (plain / repeats / low complexity / tiny), a read length, a cost model (stock -n or a
random error profile with -X) and read noise (substitutions, indels, N's), maps a few thousand reads both ways and stops at
the first difference.  usage: python tests/fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import capi          # noqa: E402
import orc           # noqa: E402
import simulate as S  # noqa: E402


def genome(kind, rng):
    if kind == "plain":
        return [("p%d" % i, S.make_contig(int(n), rng, [(300, 340)] if i == 0 else [])) for i, n in enumerate(rng.integers(20000, 400000, 3))]
    if kind == "repeats":
        a = S.make_contig(120000, rng, [])
        for _ in range(12):                                  # tandem and dispersed copies, some diverged
            ln = int(rng.integers(40, 3000)); s = int(rng.integers(0, a.size - ln)); d = int(rng.integers(0, a.size - ln))
            seg = a[s:s + ln].copy()
            mut = rng.random(ln) < rng.choice([0.0, 0.01, 0.05])
            seg[mut] = S.BASES[rng.integers(0, 4, int(mut.sum()))]
            a[d:d + ln] = seg
        return [("r0", a), ("r1", S.make_contig(30000, rng, [(10, 60), (15000, 15030)]))]
    if kind == "lowcomplexity":
        a = S.make_contig(60000, rng, [], at=0.45)
        for _ in range(20):
            ln = int(rng.integers(20, 400)); d = int(rng.integers(0, a.size - ln))
            unit = S.BASES[rng.integers(0, 4, int(rng.integers(1, 5)))]
            a[d:d + ln] = np.resize(unit, ln)
        return [("l0", a)]
    return [("t0", S.make_contig(int(rng.integers(300, 3000)), rng, [], softmask_frac=0.0))]      # tiny: every k-mer repeats


def run(n_cases, seed):
    """returns the number of identical cases; raises SystemExit(1) at the first difference"""
    rng = np.random.default_rng(seed)
    work = tempfile.mkdtemp(prefix="psfuzz_")
    t0 = time.time()
    for case in range(n_cases):
        kind = rng.choice(["plain", "repeats", "lowcomplexity", "tiny"], p=[0.35, 0.3, 0.2, 0.15])
        g = genome(kind, rng)
        fa = os.path.join(work, "g%d.fa" % case)
        S.write_fasta(fa, g)
        L = int(rng.choice([20, 28, 36, 50, 51, 64, 65, 75, 100, 150]))
        L = min(L, min(a.size for _, a in g) - 4)
        mixed = rng.random() < 0.25
        n_reads = 1500 if L > 75 else 3000
        P = S.EXAMPLE_PROFILE.copy()
        if rng.random() < 0.7:
            P[3, 1] = rng.choice([0.02, 0.12, 0.3]); P[3, 3] = 1.0 - P[3, 1] - P[3, 0] - P[3, 2]
        sim = S.simulate_reads(g, n_reads=n_reads, read_len=L, min_len=max(17, L - 20) if mixed else None, seed=int(rng.integers(1 << 30)),
                               bound=float(rng.choice([0.0, 0.6, 1.0])), profile=P, indel_scale=float(rng.choice([0, 30, 200])),
                               n_frac=float(rng.choice([0, 0.002, 0.02])))
        fq = os.path.join(work, "r%d.fq" % case)
        S.write_fastq(fq, sim)
        ctx = capi.Ctx.build(fa, device=0)
        oix = orc.Index.from_fasta(fa)
        if rng.random() < 0.45:
            n = str(rng.choice(["0.04", "0.02", "0", "1", "2", "4"]))
            ctx.set_stock(n); opt = orc.stock_opt(n); what = "stock -n " + n
        else:
            x = int(rng.choice([-1, 1, 2, 3]))
            ins, dele = float(rng.choice([0.0, 2.1e-5, 1e-3])), float(rng.choice([0.0, 5.9e-4, 1e-2]))
            ctx.set_profile(P, ins, dele, x); opt = orc.profile_opt(P, ins, dele, x); what = "profile -X %d T>C %.2f ins %g del %g" % (x, P[3, 1], ins, dele)
        if rng.random() < 0.3:                                # small tiers: exercise in-launch growth and the larger tiers
            ctx.set_tiers([int(rng.choice([64, 512])), 4096, 2000064], [2, 64, 65536], 0)
        b = ctx.batch_from_fastq(fq)
        b.run(threads=8)
        gsam, osam, osai = fq + ".g.sam", fq + ".o.sam", fq + ".o.sai"
        b.write_sam(gsam)
        oix.map_fastq(opt, fq, osam, sai_out=osai, n_threads=16)
        gl = [l for l in open(gsam) if not l.startswith("@PG")]
        ol = [l for l in open(osam) if not l.startswith("@PG")]
        bad = [i for i, (x_, y_) in enumerate(zip(gl, ol)) if x_ != y_]
        tm = b.timing()
        print("case %d: %s genome, L=%d%s, %s: %d reads, tiers re-run %d/%d -> %s" % (case, kind, L, " mixed" if mixed else "", what, n_reads,
              tm["n_overflow_tier1"], tm["n_overflow_tier2"], "identical" if not bad and len(gl) == len(ol) else "DIFFERENT"), flush=True)
        if bad or len(gl) != len(ol):
            print("first difference at line", bad[0] if bad else min(len(gl), len(ol)))
            print("GPU   :", gl[bad[0]] if bad else "")
            print("oracle:", ol[bad[0]] if bad else "")
            print("kept:", fa, fq)
            sys.exit(1)
        n_aln = b.n_aln()
        sai = orc.read_sai(osai)
        assert n_aln.tolist() == [len(x_) for x_ in sai], "hit-list lengths differ"
        b.free(); ctx.close()
        for p in (gsam, osam, osai, fq, fa):
            os.remove(p)
    print("all %d cases identical (%.0f s)" % (n_cases, time.time() - t0))
    return n_cases


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
