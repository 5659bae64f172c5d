"""CPU tier: the read parser behind ps_map / ps_batch_from_fastq (host code, no GPU call): the same ReadSet whether the file
is parsed whole on one thread, whole on many threads (cut at verified record starts) or streamed in small windows -- for
four-line FASTQ, wrapped (multi-line) FASTQ, quality strings that start with '@' or '+', CRLF line ends, FASTA reads and
input without a final newline.  (cut_records could split a wrapped record; ps_map read the file whole.)"""
import os

import numpy as np
import pytest

import capi


def _records(n, rng, lo=30, hi=120):
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        seq = "".join("ACGTN"[int(c)] for c in rng.choice(5, L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        q = rng.integers(33, 74, L)
        if i % 7 == 0:
            q[0] = ord("@")                    # a quality line that looks like a header
        if i % 11 == 0:
            q[0] = ord("+")
        out.append(("read%d/1" % i if i % 3 == 0 else "read%d extra words" % i, seq, "".join(chr(int(c)) for c in q)))
    return out


def _write(path, recs, wrap=0, crlf=False, fasta=False, final_newline=True):
    nl = "\r\n" if crlf else "\n"
    def lines(s):
        return [s] if not wrap else [s[a:a + wrap] for a in range(0, len(s), wrap)]
    parts = []
    for name, seq, q in recs:
        if fasta:
            parts += [">" + name] + lines(seq)
        else:
            parts += ["@" + name] + lines(seq) + ["+"] + lines(q)
    text = nl.join(parts) + (nl if final_newline else "")
    with open(path, "w", newline="") as f:
        f.write(text)


@pytest.mark.parametrize("kind", ["plain", "wrapped", "crlf", "fasta", "nofinal", "wrapped17"])
def test_parser_same_result_however_it_is_cut(tmp_path, kind):
    rng = np.random.default_rng(5)
    recs = _records(30000, rng)                # ~6 MB: above the 1 MB below which no thread cut is made
    p = str(tmp_path / ("r." + kind))
    _write(p, recs, wrap={"wrapped": 40, "wrapped17": 17}.get(kind, 0), crlf=kind == "crlf", fasta=kind == "fasta",
           final_newline=kind != "nofinal")
    ref = capi.ps_parse_check(p, 1, 0)
    assert ref[0] == len(recs) and ref[1] == sum(len(r[1]) for r in recs) and ref[3] == 1
    assert capi.ps_parse_check(p, 8, 0)[:3] == ref[:3]
    for window in (4096, 100_000, 1_000_000):
        got = capi.ps_parse_check(p, 3, window)
        assert got[:3] == ref[:3], (kind, window)
        if window < 1_000_000:
            assert got[3] > 4                  # really streamed in pieces


def test_parser_small_inputs(tmp_path):
    p = str(tmp_path / "one.fq")
    open(p, "w").write("@a\nACGT\n+\nIIII\n")
    assert capi.ps_parse_check(p, 4, 0)[:2] == (1, 4) and capi.ps_parse_check(p, 4, 4096)[:2] == (1, 4)
    open(p, "w").write("")
    assert capi.ps_parse_check(p, 4, 0)[0] == 0 and capi.ps_parse_check(p, 4, 4096)[0] == 0
    with pytest.raises(capi.PsError):
        capi.ps_parse_check(str(tmp_path / "missing.fq"), 1, 0)
