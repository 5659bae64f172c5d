#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.  SELF-GENERATED: the reference tree holds no
aligner source, tests or SAM fixtures, so these vectors come from this
repository's CPU oracle (oracle/ps_oracle.c) and pin the GPU path -- and the oracle itself against
drift -- not the PARA-suite_aligner binary.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "para-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc          # noqa: E402
import simulate as S  # noqa: E402


def main():
    rng = np.random.default_rng(20261004)
    a = S.make_contig(9000, rng, [(0, 40), (4000, 4120)], softmask_frac=0.3)
    b = S.make_contig(6000, rng, [], softmask_frac=0.0)
    unit = b[1000:1031].copy()
    for t in range(8):
        b[2000 + 31 * t:2000 + 31 * (t + 1)] = unit           # tandem repeat
    c = S.make_contig(5000, rng, [(4960, 5000)], softmask_frac=0.0)
    c[1000:1400] = a[6000:6400]                                # duplication across contigs
    g = [("gA", a), ("gB", b), ("gC", c)]
    fa = os.path.join(HERE, "golden.fa")
    S.write_fasta(fa, g, width=60)
    sim = S.simulate_reads(g, 160, 50, seed=4242, indel_scale=60, n_frac=0.004)
    sim2 = S.simulate_reads(g, 60, 75, min_len=30, seed=4243, indel_scale=60)
    fq = os.path.join(HERE, "golden.fq")
    S.write_fastq(fq, sim)
    with open(fq, "ab") as f, open(os.path.join(HERE, "_tmp.fq"), "wb"):
        pass
    tmp = os.path.join(HERE, "_tmp.fq")
    S.write_fastq(tmp, sim2, names=["var%d" % i for i in range(60)])
    with open(fq, "ab") as f:
        f.write(open(tmp, "rb").read())
        f.write(b"@allN\n" + b"N" * 36 + b"\n+\n" + b"I" * 36 + b"\n@polyA/1\n" + b"A" * 50 + b"\n+\n" + b"I" * 50 + b"\n")
    os.remove(tmp)
    # profile files in the format ErrorProfiling writes (ErrorProfiling.java:504-531,545-591)
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    with open(os.path.join(HERE, "golden.errorprofile"), "w") as f:
        for row in P:
            f.write("".join(repr(float(v)) + "\t" for v in row) + "\n")
    with open(os.path.join(HERE, "golden.indelprofile"), "w") as f:
        f.write("2.1E-5\t5.9E-4")
    ix = orc.Index.from_fasta(fa)
    ix.map_fastq(orc.stock_opt("2"), fq, os.path.join(HERE, "golden.stock_n2.sam"), sai_out=os.path.join(HERE, "_s.sai"))
    sai = orc.read_sai(os.path.join(HERE, "_s.sai"))
    os.remove(os.path.join(HERE, "_s.sai"))
    json.dump([[[int(x[f]) for f in ("k", "l", "n_mm", "n_gapo", "n_gape", "n_ins", "n_del", "score")] for x in r] for r in sai],
              open(os.path.join(HERE, "golden.stock_n2.intervals.json"), "w"))
    ix.map_fastq(orc.profile_opt(P, 2.1e-5, 5.9e-4, -1), fq, os.path.join(HERE, "golden.profile_x-1.sam"))
    print("written", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
