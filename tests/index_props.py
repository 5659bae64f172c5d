"""Size-independent properties of an FM index, computed FROM THE TEXT (never from the index's own BWT): what a wrong suffix
order, a wrong high bit or a wrong N-fill would break.  Used on the GPU at 2.2 Gbp (tests/test_gpu_rows33.py) and on the CPU
with a deliberately damaged suffix array (tests/test_index_props_cpu.py: the check must fail there).

Text convention: T = fwd || revcomp(fwd), n = 2 * l_pac symbols 0..3, T$ has rows 0..n,
SA[0] = n (the empty suffix; '$' sorts first), BWT[row] = T[SA[row] - 1] ('$' for the row whose SA is 0: `primary`)."""
import numpy as np


class Text:
    """T = fwd || revcomp(fwd) addressed without materialising the second half"""

    def __init__(self, fwd_codes):
        self.fwd = np.ascontiguousarray(fwd_codes, dtype=np.uint8)
        self.l_pac = int(self.fwd.size)
        self.n = 2 * self.l_pac

    def at(self, pos):
        """symbols at positions pos (int64 array); -1 for pos >= n (the '$' and beyond)"""
        pos = np.asarray(pos, dtype=np.int64)
        out = np.full(pos.shape, -1, dtype=np.int16)
        a = pos < self.l_pac
        out[a] = self.fwd[pos[a]]
        b = (~a) & (pos < self.n)
        out[b] = 3 - self.fwd[self.n - 1 - pos[b]].astype(np.int16)
        return out


def suffix_less(text, p, q, step=64, max_len=1 << 20):
    """elementwise: suffix T[p..] < suffix T[q..]  (p != q); also returns the common prefix lengths"""
    p = np.asarray(p, dtype=np.int64).copy()
    q = np.asarray(q, dtype=np.int64).copy()
    res = np.zeros(p.size, dtype=bool)
    lcp = np.zeros(p.size, dtype=np.int64)
    todo = np.arange(p.size)
    off = 0
    ar = np.arange(step, dtype=np.int64)[None, :]
    while todo.size and off < max_len:
        a = text.at(p[todo, None] + off + ar)
        b = text.at(q[todo, None] + off + ar)
        ne = a != b
        any_ne = ne.any(1)
        first = np.argmax(ne, axis=1)
        idx = todo[any_ne]
        fa, fb = a[any_ne, first[any_ne]], b[any_ne, first[any_ne]]
        res[idx] = fa < fb                     # -1 ('$' / beyond the end) sorts first
        lcp[idx] = off + first[any_ne]
        todo = todo[~any_ne]
        off += step
    if todo.size:
        raise AssertionError("suffixes equal for %d symbols" % max_len)
    return res, lcp


def check_index(text, rows, sa_lookup, bwt_lookup, primary, L2):
    """rows: int64 array of rows r with 1 <= r <= n - 1; checks the pairs (r, r+1).  Returns a dict of what was seen; raises
    AssertionError with the first violation."""
    rows = np.asarray(rows, dtype=np.int64)
    assert rows.min() >= 1 and rows.max() + 1 <= text.n
    sa0 = np.asarray(sa_lookup(rows), dtype=np.int64)
    sa1 = np.asarray(sa_lookup(rows + 1), dtype=np.int64)
    assert (sa0 >= 0).all() and (sa0 < text.n).all() and (sa1 >= 0).all() and (sa1 < text.n).all(), "SA value outside the text"
    assert (sa0 != sa1).all(), "two rows with the same suffix"
    less, lcp = suffix_less(text, sa0, sa1)
    bad = np.flatnonzero(~less)
    assert bad.size == 0, "suffix order violated at rows %s (SA %s vs %s, common prefix %s)" % (
        rows[bad[:3]].tolist(), sa0[bad[:3]].tolist(), sa1[bad[:3]].tolist(), lcp[bad[:3]].tolist())
    if bwt_lookup is not None:                 # BWT[r] == T[SA[r] - 1]
        sym = np.asarray(bwt_lookup(rows), dtype=np.int16)
        exp = text.at(np.where(sa0 > 0, sa0 - 1, 0))
        ok = (sym == exp) | (sa0 == 0)
        assert ok.all(), "BWT symbol differs from the text at rows %s" % rows[np.flatnonzero(~ok)[:3]].tolist()
        assert ((sa0 == 0) == (rows == primary)).all(), "primary row is not the row of the whole text"
    if L2 is not None:                         # C array against the base counts of T
        cnt = np.bincount(text.fwd, minlength=4)[:4].astype(np.int64)
        tot = cnt + cnt[::-1]                  # revcomp(fwd) holds 3 - c
        assert [int(L2[c + 1]) - int(L2[c]) for c in range(4)] == tot.tolist(), "L2 does not match the base counts of the text"
        first = text.at(sa0)
        assert (first == np.searchsorted(np.asarray(L2[1:5], dtype=np.int64), rows - 1, side="right")).all(), \
            "a row's suffix starts with the wrong symbol for its L2 range"
    return dict(pairs=int(rows.size), max_lcp=int(lcp.max()), long_lcp=int((lcp >= 100).sum()), above_2_32=int((rows >= 2 ** 32).sum()))
