"""CPU tier: the device core (para-suite_amd/csrc/ps_core.h) executed lane by lane on the host
(tests/hostsim) against the oracle.  This exercises the exact kernel state machine -- Occ blocks,
width chains, per-lane stack with score buckets, SA walk, banded DP -- without a GPU.  It is a test
harness, not a product path: libparasuite_hip.so has no CPU implementation."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc
import simulate as S

HERE = os.path.dirname(os.path.abspath(__file__))
ALNREC = np.dtype([("k", "<u8"), ("l", "<u8"), ("score", "<u2"), ("units", "<u2"), ("n_mm", "u1"), ("n_gapo", "u1"),
                   ("n_gape", "u1"), ("n_ins", "u1"), ("n_del", "u1"), ("pad", "u1", 7)])


@pytest.fixture(scope="module")
def hs():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostsim")])
    H = C.CDLL(os.path.join(HERE, "hostsim", "libhostsim.so"))
    H.hs_index_new.restype = C.c_void_p
    H.hs_index_new.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
    H.hs_index_free.argtypes = [C.c_void_p]
    H.hs_occ.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    H.hs_occ.restype = C.c_uint32
    H.hs_sa.argtypes = [C.c_void_p, C.c_uint64]
    H.hs_sa.restype = C.c_uint64
    H.hs_sizeof_model.restype = C.c_size_t
    H.hs_sizeof_alnrec.restype = C.c_size_t
    H.hs_model_stock.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    H.hs_model_profile.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p]
    H.hs_aln.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
    H.hs_set_est.argtypes = [C.c_void_p]
    H.hs_index_jump.argtypes = [C.c_void_p, C.c_int]
    H.hs_jump_slot.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    H.hs_banded.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_int]
    assert H.hs_sizeof_alnrec() == ALNREC.itemsize
    return H


@pytest.fixture(scope="module")
def sim_index(hs, example):
    ix = example["orc_index"]
    bw, sa, pac = ix.bwt_syms(), ix.sa_samples(), ix.pac()
    h = hs.hs_index_new(bw.ctypes.data, ix.seq_len, ix.primary, sa.ctypes.data, sa.size, 32, pac.ctypes.data, ix.l_pac)
    assert hs.hs_index_jump(h, -1) >= 5                     # with the jump table, as deep as the product takes it for this text
    yield h
    hs.hs_index_free(h)


@pytest.fixture(scope="module")
def sim_index_plain(hs, example):
    """the same index without a jump table: every step through the Occ blocks (what the counting kernel and the wide tier do)"""
    ix = example["orc_index"]
    bw, sa, pac = ix.bwt_syms(), ix.sa_samples(), ix.pac()
    h = hs.hs_index_new(bw.ctypes.data, ix.seq_len, ix.primary, sa.ctypes.data, sa.size, 32, pac.ctypes.data, ix.l_pac)
    yield h
    hs.hs_index_free(h)


def test_jump_table_slots(hs, sim_index, example):
    """a slot holds Occ(k-1, .) and Occ(l, .) of its string's interval (the string's symbols in the order the backward search
    consumes them), all zero where the string does not occur -- against a plain backward search with the oracle's Occ"""
    ix = example["orc_index"]
    L2 = [int(x) for x in ix.L2]
    levels = hs.hs_index_jump(sim_index, 11)                # deeper than the text warrants: strings that do not occur
    rng = np.random.default_rng(5)
    out = np.zeros(8, dtype=np.uint32)
    seen_empty = 0
    for j in range(600):
        d = int(rng.integers(0, levels)) if j < 300 else levels - 1
        syms = [int(x) for x in rng.integers(0, 4, d)]
        k, l = 0, ix.seq_len
        for c in syms:
            k, l = L2[c] + ix.occ(k - 1, c) + 1, L2[c] + ix.occ(l, c)
            if k > l:
                break
        hs.hs_jump_slot(sim_index, d, _pad(syms), out.ctypes.data)
        if k > l:
            seen_empty += 1
            assert not out.any()
        else:
            exp = []
            for c in range(4):
                exp += [ix.occ(k - 1, c), ix.occ(l, c)]
            assert out.tolist() == exp, (d, syms)
    assert seen_empty > 0
    hs.hs_index_jump(sim_index, -1)


def _pad(syms):
    """index of the whole string (the search above stops at the first empty prefix: every extension of it is empty too)"""
    v = 0
    for c in syms:
        v = v * 4 + c
    return v


def test_occ_blocks_and_sa_walk(hs, sim_index, example):
    ix = example["orc_index"]
    rng = np.random.default_rng(0)
    rows = list(rng.integers(-1, ix.seq_len + 1, 1500)) + [-1, 0, ix.primary - 1, ix.primary, ix.primary + 1, ix.seq_len - 1, ix.seq_len]
    rows += [191, 192, 193, 383, 384]                      # block edges
    for k in rows:
        for c in range(4):
            assert hs.hs_occ(sim_index, int(k), c) == ix.occ(int(k), c), (k, c)
    for k in list(rng.integers(0, ix.seq_len + 1, 1500)) + [0, ix.primary, ix.seq_len]:
        assert hs.hs_sa(sim_index, int(k)) == ix.sa(int(k))


def _run(hs, h, model, codes, n_lanes=64, pool_cap=4096, aln_cap=64, wide=0, n_big=0, est=None):
    n, L = codes.shape
    if est is not None:
        est = np.ascontiguousarray(est, dtype=np.uint8)
        assert est.shape == (n,)
        hs.hs_set_est(est.ctypes.data)          # for the call below only
    alns = np.zeros((n, aln_cap), dtype=ALNREC)
    n_aln = np.zeros(n, dtype=np.int32)
    status = np.zeros(n, dtype=np.uint8)
    ks = np.zeros(8, dtype=np.uint64)
    cc = np.ascontiguousarray(codes)
    hs.hs_aln(h, model, n, L, cc.ctypes.data, n_lanes, pool_cap, aln_cap, wide, n_big, None, None, None, alns.ctypes.data,
              n_aln.ctypes.data, status.ctypes.data, ks.ctypes.data)
    return alns, n_aln, status, ks


def _compare(hs, h, ix, opt, model, codes, **kw):
    alns, n_aln, status, ks = _run(hs, h, model, codes, **kw)
    for r in range(codes.shape[0]):
        n, ref = ix.aln_one(opt, codes[r], cap=64)
        assert status[r] == 0
        got = [tuple(int(a[f]) for f in ("k", "l", "n_mm", "n_gapo", "n_gape", "n_ins", "n_del", "score", "units")) for a in alns[r, :n_aln[r]]]
        exp = [tuple(a[f] for f in ("k", "l", "n_mm", "n_gapo", "n_gape", "n_ins", "n_del", "score", "units")) for a in ref]
        assert n == n_aln[r] and got == exp, r
    return ks


@pytest.mark.parametrize("n_arg", ["0.04", "2", "0", "4"])
def test_lane_machine_stock(hs, sim_index, example, n_arg):
    sim = S.simulate_reads(example["genome"], 500, 50, seed=3, indel_scale=40, n_frac=0.002)
    model = (C.c_uint8 * hs.hs_sizeof_model())()
    assert hs.hs_model_stock(n_arg.encode(), 50, model) == 0
    ks = _compare(hs, sim_index, example["orc_index"], orc.stock_opt(n_arg), model, sim["codes"])
    assert ks[0] > 0


@pytest.mark.parametrize("x", [-1, 2, 1])
def test_lane_machine_profile(hs, sim_index, example, x):
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    Pc = np.ascontiguousarray(P.reshape(16))
    sim = S.simulate_reads(example["genome"], 400, 50, seed=4, indel_scale=40)
    model = (C.c_uint8 * hs.hs_sizeof_model())()
    assert hs.hs_model_profile(Pc.ctypes.data, 2.1e-5, 5.9e-4, x, 50, model) == 0
    _compare(hs, sim_index, example["orc_index"], orc.profile_opt(P, 2.1e-5, 5.9e-4, x), model, sim["codes"])


@pytest.mark.parametrize("L", [14, 20, 32, 33, 36, 75, 101])
def test_lane_machine_lengths(hs, sim_index, example, L):
    sim = S.simulate_reads(example["genome"], 150, L, seed=5 + L, indel_scale=40)
    model = (C.c_uint8 * hs.hs_sizeof_model())()
    assert hs.hs_model_stock(b"0.04", L, model) == 0
    _compare(hs, sim_index, example["orc_index"], orc.stock_opt("0.04"), model, sim["codes"])


def test_lane_reuse_and_few_lanes(hs, sim_index, example):
    """a lane maps many reads in turn (static read assignment r = lane + j*n_lanes)"""
    sim = S.simulate_reads(example["genome"], 300, 50, seed=17, indel_scale=40)
    model = (C.c_uint8 * hs.hs_sizeof_model())()
    assert hs.hs_model_stock(b"0.04", 50, model) == 0
    _compare(hs, sim_index, example["orc_index"], orc.stock_opt("0.04"), model, sim["codes"], n_lanes=3)


@pytest.mark.parametrize("n_arg,L", [("0.04", 50), ("2", 32), ("0.04", 75)])
def test_lane_machine_without_jump_table(hs, sim_index_plain, example, n_arg, L):
    sim = S.simulate_reads(example["genome"], 300, L, seed=29 + L, indel_scale=40, n_frac=0.002)
    model = (C.c_uint8 * hs.hs_sizeof_model())()
    assert hs.hs_model_stock(n_arg.encode(), L, model) == 0
    _compare(hs, sim_index_plain, example["orc_index"], orc.stock_opt(n_arg), model, sim["codes"])


@pytest.mark.parametrize("L", [6, 8, 11])
def test_hits_inside_the_jump_table_levels(hs, sim_index, example, L):
    """reads shorter than the table is deep end their search on an entry that carries (string index, width): the hit takes the
    interval back out of the parent's slot"""
    levels = hs.hs_index_jump(sim_index, 12)
    try:
        assert levels == 12
        sim = S.simulate_reads(example["genome"], 300, L, seed=31 + L, indel_scale=40)
        model = (C.c_uint8 * hs.hs_sizeof_model())()
        assert hs.hs_model_stock(b"1", L, model) == 0
        _compare(hs, sim_index, example["orc_index"], orc.stock_opt("1"), model, sim["codes"])
    finally:
        hs.hs_index_jump(sim_index, -1)


def test_wide_stack_variant(hs, sim_index, example):
    """the last tier's 32-byte entries / free list / global heads give the same hits as the packed 16-byte stack"""
    sim = S.simulate_reads(example["genome"], 200, 50, seed=19, indel_scale=40)
    model = (C.c_uint8 * hs.hs_sizeof_model())()
    assert hs.hs_model_stock(b"0.04", 50, model) == 0
    _compare(hs, sim_index, example["orc_index"], orc.stock_opt("0.04"), model, sim["codes"], wide=1, pool_cap=4096)


def test_stack_grows_into_large_slot(hs, sim_index, example):
    """a read that outgrows its private stack slice continues in a large slot (entries copied, indices kept)"""
    sim = S.simulate_reads(example["genome"], 200, 50, seed=23, indel_scale=40)
    model = (C.c_uint8 * hs.hs_sizeof_model())()
    assert hs.hs_model_stock(b"0.04", 50, model) == 0
    _compare(hs, sim_index, example["orc_index"], orc.stock_opt("0.04"), model, sim["codes"], pool_cap=24, n_big=1000)
    _, _, status, _ = _run(hs, sim_index, model, sim["codes"], pool_cap=24, n_big=3)
    assert (status == 1).any() and (status == 0).any()       # slots exhausted: the rest is reported for the next tier


def test_pool_overflow_is_reported(hs, sim_index, example):
    sim = S.simulate_reads(example["genome"], 100, 50, seed=18)
    model = (C.c_uint8 * hs.hs_sizeof_model())()
    assert hs.hs_model_stock(b"0.04", 50, model) == 0
    _, _, status, _ = _run(hs, sim_index, model, sim["codes"], pool_cap=8)
    assert (status == 1).any()                               # RS_OVERFLOW_POOL -> the host escalates the tier


def test_banded_dp_matches_oracle(hs, sim_index, example):
    fwd = example["orc_index"].forward_codes()
    rng = np.random.default_rng(6)
    for trial in range(40):
        p = int(rng.integers(20000, 200000))
        tlen = 50 + int(rng.integers(-3, 4))
        t = fwd[p:p + tlen]
        q = list(fwd[p:p + 50])
        if tlen > 50:
            q = list(fwd[p:p + 20]) + list(fwd[p + 20 + (tlen - 50):p + tlen])
        elif tlen < 50:
            ins = rng.integers(0, 4, 50 - tlen).tolist()
            q = list(fwd[p:p + 20]) + ins + list(fwd[p + 20:p + tlen])
        q = np.array(q[:50] + [0] * (50 - len(q[:50])), dtype=np.uint8)
        w = max(50, int(abs(tlen - 50) * 1.5))
        cig = (C.c_uint32 * 16)()
        n = hs.hs_banded(sim_index, 50, q.ctypes.data, p, tlen, w, cig, 16)
        got = [(c >> 4, "MIDS"[c & 0xF]) for c in cig[:n]]
        assert got == orc.ksw_global(q, t, w), trial


def test_rows_above_32_bits_round_trip(hs):
    """33-bit rows (hg19 has 6.27e9): stack entries, block addressing, width saturation, sampled SA with its bit-32 plane."""
    hs.hs_unit_rows33.restype = C.c_int
    assert hs.hs_unit_rows33() == 0


@pytest.mark.parametrize("x", [-1, 1])
def test_estimated_best_score_never_shows_in_the_hits(hs, sim_index, example, x):
    """ps_narrow.h (nt_tail): with an estimate of a read's best score the lane leaves out children that can only matter if the best
    hit is worse, and starts the read over without one when the estimate fails.  Whatever the estimate -- exact, too low, too high,
    zero, random -- hit lists and status equal the run without one (which test_lane_machine_profile compares with the oracle);
    exact estimates store fewer entries, failed ones cost a second search."""
    P = S.EXAMPLE_PROFILE.copy()
    P[3, 1], P[3, 3] = 0.12, 0.87
    Pc = np.ascontiguousarray(P.reshape(16))
    sim = S.simulate_reads(example["genome"], 300, 50, seed=41, indel_scale=40, n_frac=0.002)
    model = (C.c_uint8 * hs.hs_sizeof_model())()
    assert hs.hs_model_profile(Pc.ctypes.data, 2.1e-5, 5.9e-4, x, 50, model) == 0
    ks0 = _compare(hs, sim_index, example["orc_index"], orc.profile_opt(P, 2.1e-5, 5.9e-4, x), model, sim["codes"])
    a0, n0, s0, _ = _run(hs, sim_index, model, sim["codes"])
    best = np.where(n0 > 0, a0["units"][:, 0], 255).astype(np.int64)
    rng = np.random.default_rng(9)
    pushes = {}
    for tag, est in (("exact", best), ("low", np.maximum(best - 8, 0)), ("high", np.minimum(best + 5, 255)), ("zero", np.zeros_like(best)),
                     ("random", rng.integers(0, 40, best.size))):
        a1, n1, s1, ks = _run(hs, sim_index, model, sim["codes"], est=np.clip(est, 0, 255))
        assert np.array_equal(n1, n0) and np.array_equal(s1, s0), tag
        for r in range(best.size):
            assert a1[r, :n1[r]].tobytes() == a0[r, :n0[r]].tobytes(), (tag, r)
        pushes[tag] = int(ks[3])
    assert pushes["exact"] <= int(ks0[3]) and pushes["high"] >= pushes["exact"], (pushes, int(ks0[3]))
    if x == -1:                                  # a budget of several differences: room between the estimate and the budget (-X 1 has none)
        assert pushes["exact"] < 0.9 * int(ks0[3]), (pushes, int(ks0[3]))
        assert pushes["zero"] > pushes["exact"], pushes
