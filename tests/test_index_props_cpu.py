"""CPU tier: the text-based index properties (tests/index_props.py) hold for a correct suffix array and FAIL for damaged ones --
ties left unsorted (what skipping a round of the builder's tie-only prefix doubling would produce), two neighbours of a long
tie group swapped, one wrong position bit, a wrong C array.  The suffix array here is built by plain numpy prefix doubling:
nothing of the product or of the oracle."""
import numpy as np
import pytest

import index_props as IP


def _suffix_array(T):
    """SA of T$ by prefix doubling (numpy); row 0 = the empty suffix"""
    n = T.size
    rank = np.concatenate([T.astype(np.int64) + 1, [0]])           # '$' = 0
    k = 1
    while True:
        r2 = np.concatenate([rank[k:], np.zeros(k, dtype=np.int64)])
        key = rank * (n + 2) + r2
        sa = np.argsort(key, kind="stable")
        ks = key[sa]
        new = np.empty(n + 1, dtype=np.int64)
        new[sa] = np.concatenate([[0], np.cumsum(ks[1:] != ks[:-1])])
        rank = new
        if rank.max() == n:
            return sa
        k *= 2


@pytest.fixture(scope="module")
def small():
    rng = np.random.default_rng(12)
    fwd = rng.integers(0, 4, 30000).astype(np.uint8)
    for _ in range(6):                                              # dispersed copies: long ties, some diverged by a base
        ln = int(rng.integers(300, 2000))
        s = int(rng.integers(0, fwd.size - ln))
        d = int(rng.integers(0, fwd.size - ln))
        seg = fwd[s:s + ln].copy()
        if rng.random() < 0.5:
            seg[ln // 2] = (seg[ln // 2] + 1) & 3
        fwd[d:d + ln] = seg
    text = IP.Text(fwd)
    T = text.at(np.arange(text.n)).astype(np.uint8)
    sa = _suffix_array(T)
    bwt = np.where(sa > 0, T[np.maximum(sa, 1) - 1], 255).astype(np.int16)
    cnt = np.bincount(T, minlength=4)
    L2 = np.concatenate([[0], np.cumsum(cnt)])
    return dict(text=text, T=T, sa=sa, bwt=bwt, primary=int(np.flatnonzero(sa == 0)[0]), L2=L2)


def _check(small, sa, rows=None):
    n = small["text"].n
    rows = np.arange(1, n) if rows is None else rows
    return IP.check_index(small["text"], rows, lambda r: sa[r], lambda r: small["bwt"][r], small["primary"], small["L2"])


def test_correct_index_passes(small):
    seen = _check(small, small["sa"])
    assert seen["pairs"] == small["text"].n - 1 and seen["long_lcp"] > 1000 and seen["max_lcp"] >= 300


def test_unsorted_ties_are_caught(small):
    """suffixes ordered by their first 64 symbols only, ties left in text order: a doubling round that never ran"""
    T, n = small["T"], small["text"].n
    pad = np.concatenate([T.astype(np.int16), np.full(64, -1, dtype=np.int16)])
    win = np.lib.stride_tricks.sliding_window_view(pad, 64)[:n + 1]
    order = np.lexsort(win.T[::-1])                                   # stable: equal keys stay in text order
    assert (order != small["sa"]).any()
    with pytest.raises(AssertionError, match="suffix order violated"):
        _check(small, order)


def test_swapped_neighbours_in_a_tie_group_are_caught(small):
    sa = small["sa"].copy()
    text = small["text"]
    rows = np.arange(1, text.n)
    _, lcp = IP.suffix_less(text, sa[rows], sa[rows + 1])
    r = int(rows[np.argmax(lcp)])                                     # the longest tie: what only the last rounds separate
    sa[r], sa[r + 1] = sa[r + 1], sa[r]
    with pytest.raises(AssertionError, match="suffix order violated"):
        _check(small, sa, rows=np.array([r]))
    with pytest.raises(AssertionError):
        _check(small, sa, rows=np.array([r - 1, r + 1]))            # the neighbours see it as well (order or BWT symbol)


def test_wrong_position_bit_is_caught(small):
    sa = small["sa"].copy()
    r = 1234
    sa[r] ^= 1 << 13                                                  # stands for a wrong bit 32 at full size
    with pytest.raises(AssertionError):
        _check(small, sa, rows=np.array([r - 1, r]))


def test_wrong_l2_is_caught(small):
    L2 = small["L2"].copy()
    L2[2] += 1
    with pytest.raises(AssertionError, match="L2"):
        IP.check_index(small["text"], np.arange(1, 2000), lambda r: small["sa"][r], None, small["primary"], L2)
