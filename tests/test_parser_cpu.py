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


def test_pieces_by_demand_and_the_end_of_the_input(tmp_path, monkeypatch):
    """ps_map's parser hands a piece over when the GPU worker waits (here: always) and the piece holds a minimum, lets a piece end at
    its size rather than at a window boundary, and takes a short end of the input along instead of leaving it as a piece of its
    own: the same reads whatever the cut, and no small last piece"""
    rng = np.random.default_rng(6)
    recs = _records(60000, rng)                # 10.4 MB
    p = str(tmp_path / "r.fq")
    _write(p, recs)
    size = os.path.getsize(p)
    assert 10_000_000 < size < 10_600_000
    ref = capi.ps_parse_check(p, 4, 0)
    plain = capi.ps_parse_check(p, 4, 3_000_000)                       # pieces of 3 MB: 3 + 3 + 3 + 1.4
    assert plain[:3] == ref[:3] and plain[3] == 4
    along = capi.ps_parse_check(p, 4, 5_000_000)                       # 5 + 5 + 0.4: the end is less than an eighth of a piece and is taken along
    assert along[:3] == ref[:3] and along[3] == 2
    near = capi.ps_parse_check(p, 4, size - 300_000)
    assert near[:3] == ref[:3] and near[3] == 1
    monkeypatch.setenv("PS_UNIT_MB", "1")                              # 1-MB windows (64 MB in ps_map)
    monkeypatch.setenv("PS_PARSE_CHECK_HUNGRY", "1500000")             # a waiting consumer: a piece goes out as soon as it holds 1.5 MB
    got = capi.ps_parse_check(p, 4, 8 << 20)
    assert got[:3] == ref[:3] and 5 <= got[3] <= 6, got
