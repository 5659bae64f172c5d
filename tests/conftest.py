import os
import sys

import numpy as np
import pytest

try:                       # PyTorch-ROCm brings its own copy of the HIP runtime: in a process that uses both, torch has to be imported
    import torch  # noqa: F401   # BEFORE libparasuite_hip.so is loaded (the library then binds to the runtime already there); the
except ImportError:        # other order leaves torch without a device ("no ROCm-capable device is detected")
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "para-suite_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("ps"))


@pytest.fixture(scope="session")
def example(workdir):
    """483 kb synthetic contig shaped like the reference's example genome + the oracle's index of it."""
    import orc
    import simulate as S
    g = S.example_genome()
    fa = os.path.join(workdir, "example.fa")
    S.write_fasta(fa, g)
    return dict(genome=g, fa=fa, orc_index=orc.Index.from_fasta(fa))


@pytest.fixture(scope="session")
def multi(workdir):
    """Three small contigs with N runs, IUPAC codes and a tandem repeat (contig-boundary and hole logic)."""
    import orc
    import simulate as S
    rng = np.random.default_rng(77)
    a = S.make_contig(30000, rng, [(100, 160), (9000, 9400)], softmask_frac=0.3)
    a[5000:5004] = np.frombuffer(b"RYKM", dtype=np.uint8)
    b = S.make_contig(12000, rng, [], softmask_frac=0.0)
    unit = b[2000:2037].copy()
    for t in range(12):
        b[3000 + 37 * t:3000 + 37 * (t + 1)] = unit
    c = S.make_contig(8000, rng, [(0, 50), (7950, 8000)], softmask_frac=0.0)
    c[4000:4600] = b[6000:6600]          # a 600-bp duplication across contigs
    g = [("chrA", a), ("chrB desc text", b), ("chrC", c)]
    fa = os.path.join(workdir, "multi.fa")
    with open(fa, "wb") as f:
        for name, asc in g:
            f.write(b">" + name.encode() + b"\n")
            for i in range(0, asc.size, 60):
                f.write(asc[i:i + 60].tobytes() + b"\n")
    contigs = [(n.split()[0], x) for n, x in g]
    return dict(genome=contigs, fa=fa, orc_index=orc.Index.from_fasta(fa))


def sam_records(path):
    """alignment lines only (headers differ by @PG)"""
    return [l for l in open(path).read().split("\n") if l and not l.startswith("@")]


def sam_sq(path):
    return [l for l in open(path).read().split("\n") if l.startswith("@SQ")]


@pytest.fixture(scope="session")
def mid(workdir):
    """8 Mbp, four contigs: deep enough that SA intervals stay non-empty for ~11 levels (more search per read)"""
    import orc
    import simulate as S
    g = S.big_genome(8_000_000, 4, seed=0x5EED0007)
    fa = os.path.join(workdir, "mid.fa")
    S.write_fasta(fa, g)
    return dict(genome=g, fa=fa, orc_index=orc.Index.from_fasta(fa))
