"""An independent check of what the oracle's search returns for GAPPED hits and for the PROFILE cost model (the
brute-force aligner of test_oracle.py is substitutions-only with stock costs).

`brute_best` below knows nothing of BWTs, bounds or stacks.  It is a dynamic programme over EVERY window of the genome, both strands:
the set of (text end, units spent) reachable after each read base, for every gap state, under the rules the search is meant to
implement (upstream bwt_match_gap restated in oracle/ps_oracle.c: match_gap; call sites PARAsuiteMapping.java:63-77, BWAMapping.java:51-61):

  * substitution costs from the 4x4 table (read orientation), gap open / extension costs, N = always a mismatch;
  * at most `-o` gap opens, an indel only where `-i 5` allows it (both ends of the read, shrinking with the gaps already made),
    a gap extends its own kind only, deletions sit in front of the read base being consumed;
  * budget: units <= max_units; the seed rule (`-l 32 -k 2`): a difference at one of the read's first 31 bases needs
    floor((k * unit - units so far) / cheapest cost) >= 1;
  * the D(i) lower bounds only ever prune what cannot finish inside these limits, so they do not appear here at all.

The oracle's FIRST hit must have the minimum score over everything the DP can reach, its best-score SA intervals must be exactly
the DP's optimal placements (ungapped optimum) or a subset of them (gapped: equivalent gap positions collapse), and the SAM-level
position / strand / NM / CIGAR must agree.  None of this pins the oracle to the real PARA-suite aligner (nothing can, here): it
closes the gap between "restated" and "does what the restatement says" for the least certain stage."""
import os
import sys

import numpy as np
import pytest

import orc
import simulate as S

L = 50


def _costs(opt):
    """the cost model as plain numbers, taken from the options the search is given"""
    if opt.profile:
        U = opt.unit
        sub = np.array(list(opt.sub_cost), dtype=np.int64).reshape(4, 4)             # [ref][read], read orientation
        pos = [int(v) for v in sub.flatten() if v > 0] + [opt.n_cost, opt.gapo_ins_cost, opt.gapo_del_cost, opt.gape_cost]
        diffs = opt.x_avg_mm if opt.x_avg_mm >= 0 else orc.lib().orc_cal_maxdiff(L, 0.02, 0.04)
        return dict(sub=sub, n_cost=opt.n_cost, ins_open=opt.gapo_ins_cost, del_open=opt.gapo_del_cost, ext=opt.gape_cost, c_min=max(1, min(pos)),
                    budget=diffs * U, seed_units=opt.max_seed_diff * U, score=lambda kind, g, u: u, max_gapo=opt.max_gapo)
    diffs = orc.lib().orc_cal_maxdiff(L, 0.02, opt.fnr) if opt.fnr > 0 else opt.max_diff
    sub = 1 - np.eye(4, dtype=np.int64)

    def score(kind, g, u):                   # units = differences; score = 3 mm + 11 open + 4 extension
        return opt.s_mm * u if kind == "M" else opt.s_mm * (u - g) + opt.s_gapo + opt.s_gape * (g - 1)
    return dict(sub=sub, n_cost=1, ins_open=1, del_open=1, ext=1, c_min=1, budget=diffs, seed_units=opt.max_seed_diff,
                score=score, max_gapo=min(opt.max_gapo, diffs))


def brute_best(T, read, opt):
    """-> (best score or None, set of (pos, strand, span) of every optimal placement, whether an optimal one is gapped)"""
    cm = _costs(opt)
    B, Sd, cmin, skip = cm["budget"], cm["seed_units"], cm["c_min"], opt.indel_end_skip
    n = T.size
    use_seed = L > opt.seed_len
    full = np.uint64((1 << (B + 1)) - 1)
    G = 0                                                        # longest gap that fits the budget at all
    while G < 1 + opt.max_gape and min(cm["ins_open"], cm["del_open"]) + G * cm["ext"] <= B:
        G += 1
    if cm["max_gapo"] < 1:
        G = 0
    # the deletion rule of upstream ((n_gape + n_gapo) < max_diff or few occurrences) never binds inside the budget: say so
    for g in range(1, G):
        assert g * (opt.unit if opt.profile else 1) < B or cm["del_open"] + g * cm["ext"] > B
    M0 = 0
    I = lambda g: g                                              # states: M0 | I_1..G | D_1..G | M after an insertion / deletion of g
    D = lambda g: G + g
    MI = lambda g: 2 * G + g
    MD = lambda g: 3 * G + g
    NS = 4 * G + 1
    best, places, gapped = None, set(), False
    for strand, X in ((0, T), (1, (3 - T)[::-1].copy())):
        cur = np.zeros((NS, n + 1), dtype=np.uint64)             # bit u of cur[s][t]: read[:p] can end at text position t in state s with u units
        cur[M0, :] = 1
        for p in range(L):
            seed_checked = use_seed and p <= opt.seed_len - 2 and p != L - 1
            src_ok = np.uint64((1 << (max(0, Sd - cmin) + 1)) - 1) if seed_checked else full      # units a difference may start from
            if seed_checked and Sd - cmin < 0:
                src_ok = np.uint64(0)

            def gap_ok(tmp):                                     # upstream: i >= skip + tmp and len - i >= skip + tmp, i = L - 1 - p
                return p <= L - 1 - skip - tmp and p >= skip + tmp - 1

            def pay(a, c):                                       # spend c units (a difference: only from the units the seed rule allows)
                return ((a & src_ok) << np.uint64(c)) & full
            # deletions in front of base p: text advances, the read does not
            if G and gap_ok(0):
                cur[D(1), 1:] |= pay(cur[M0, :-1], cm["del_open"])
            for g in range(1, G):
                if g - 1 < opt.max_gape and gap_ok(g):
                    cur[D(g + 1), 1:] |= pay(cur[D(g), :-1], cm["ext"])
            new = np.zeros_like(cur)
            q = int(read[p])
            cost = np.full(n, cm["n_cost"], dtype=np.int64) if q > 3 else cm["sub"][X, q]
            if q <= 3:
                cost = np.where(X == q, 0, cost)
            moves = [(M0, M0)] + [(I(g), MI(g)) for g in range(1, G + 1)] + [(D(g), MD(g)) for g in range(1, G + 1)] + \
                    [(MI(g), MI(g)) for g in range(1, G + 1)] + [(MD(g), MD(g)) for g in range(1, G + 1)]
            for c in np.unique(cost):
                m = cost == c
                for s, s2 in moves:
                    v = cur[s, :-1] if c == 0 else pay(cur[s, :-1], int(c))
                    new[s2, 1:] |= np.where(m, v, np.uint64(0))
            # insertion of base p: the read advances, the text does not
            if G and gap_ok(0):
                new[I(1), :] |= pay(cur[M0, :], cm["ins_open"])
            for g in range(1, G):
                if g - 1 < opt.max_gape and gap_ok(g):
                    new[I(g + 1), :] |= pay(cur[I(g), :], cm["ext"])
            cur = new
        finals = [("M", 0, M0, L)] + [("I", g, MI(g), L - g) for g in range(1, G + 1)] + [("D", g, MD(g), L + g) for g in range(1, G + 1)]
        for kind, g, s, span in finals:
            row = cur[s]
            for u in range(B + 1):
                ends = np.nonzero((row >> np.uint64(u)) & np.uint64(1))[0]
                if ends.size == 0:
                    continue
                sc = cm["score"](kind, g, u)
                if best is None or sc < best:
                    best, places, gapped = sc, set(), False
                if sc == best:
                    for te in ends:
                        start = int(te) - span
                        if start < 0:
                            continue
                        places.add((start if strand == 0 else n - int(te), strand, span))
                    gapped = gapped or kind != "M"
    if best is not None and not places:
        best = None
    return best, places, gapped


def _aln_places(ix, a, n):
    """(pos, strand, span) of every occurrence in the SA interval of one oracle hit"""
    span = L + a["n_del"] - a["n_ins"]
    out = set()
    for row in range(a["k"], a["l"] + 1):
        x = ix.sa(row)
        out.add((x, 1, span) if x + span <= n else (2 * n - x - span, 0, span))
    return out


def _make_reads(T, rng, n_reads, profile_mode):
    """reads with a few random substitutions, T->C conversions (profile mode) and, every third read, one short indel well inside the
    read; both strands; away from the ends of the text.  Sized to the budgets: stock 50 bp allows 3 differences; with the example
    profile a conversion costs 3, another substitution 8-9, a one-base deletion 12 and insertion 17 of the 24 units"""
    reads, truth = [], []
    for r in range(n_reads):
        start = int(rng.integers(200, T.size - 200 - L - 4))
        kind = r % 3
        glen = (1 if profile_mode else int(rng.integers(1, 4))) if kind else 0
        ref = T[start:start + L + 4].copy()
        if kind == 1:                        # deletion from the read
            at = int(rng.integers(12, L - 12))
            seq = np.concatenate([ref[:at], ref[at + glen:]])[:L]
        elif kind == 2:                      # insertion into the read
            at = int(rng.integers(12, L - 12))
            seq = np.concatenate([ref[:at], rng.integers(0, 4, glen).astype(np.uint8), ref[at:]])[:L]
        else:
            seq = ref[:L].copy()
        strand = int(rng.integers(0, 2))
        if strand:
            seq = (3 - seq)[::-1].copy()
        n_conv = n_sub = 0
        if profile_mode:                     # conversions show as T->C in the read whatever strand it came from
            n_conv = [3, 4, 5, 2, 1, 6][(r // 3) % 6] if kind == 0 else int(rng.integers(0, 3))
            n_sub = int(rng.integers(0, 2)) if kind == 0 and n_conv <= 4 else 0
            ts = np.nonzero(seq == 3)[0]
            if ts.size and n_conv:
                for j in rng.choice(ts, size=min(n_conv, ts.size), replace=False):
                    seq[j] = 1
        else:
            n_sub = int(rng.integers(0, 4)) if kind == 0 else int(rng.integers(0, 2))
        for _ in range(n_sub):
            j = int(rng.integers(0, L))
            seq[j] = (seq[j] + int(rng.integers(1, 4))) & 3
        if r % 17 == 0:
            seq[int(rng.integers(0, L))] = 4                    # an N
        reads.append(seq.astype(np.uint8)); truth.append((start, strand, kind, glen))
    return reads, truth


def _cigar_ok(cig, seq, T, pos, strand):
    """the CIGAR consumes the whole read; returns the edit distance of the alignment it describes"""
    q = seq if strand == 0 else np.where(seq > 3, 4, 3 - seq)[::-1]
    i, j, nm = pos, 0, 0
    for ln, op in cig:
        if op == 0:
            nm += int(((T[i:i + ln] != q[j:j + ln]) | (q[j:j + ln] > 3)).sum()); i += ln; j += ln
        elif op == 1:
            nm += ln; j += ln
        else:
            nm += ln; i += ln
    assert j == L
    return nm


@pytest.mark.parametrize("mode", ["stock", "profile"])
def test_first_hit_is_the_weighted_optimum(tmp_path, mode):
    rng = np.random.default_rng(20260503 + (mode == "profile"))
    g = [("c1", S.make_contig(12000, rng, [], softmask_frac=0.0))]
    g[0][1][7000:7300] = g[0][1][2000:2300]                    # a 300-bp exact repeat: several optimal placements
    fa = str(tmp_path / "g.fa")
    S.write_fasta(fa, g)
    ix = orc.Index.from_fasta(fa)
    T = ix.forward_codes()
    n = T.size
    if mode == "stock":
        opt = orc.stock_opt("0.04")
    else:
        P = S.EXAMPLE_PROFILE.copy()
        P[3, 1], P[3, 3] = 0.12, 0.87
        opt = orc.profile_opt(P, 2.1e-5, 5.9e-4, -1)
    n_reads = 96
    reads, truth = _make_reads(T, rng, n_reads, mode == "profile")
    for i in range(6):                                         # reads inside the repeat
        reads[i * 9] = T[2040 + 7 * i:2040 + 7 * i + L].copy()
        truth[i * 9] = (2040 + 7 * i, 0, 0, 0)
    sim = dict(codes=np.stack(reads), lens=np.full(n_reads, L, dtype=np.int32), quals=np.full((n_reads, L), 73, dtype=np.uint8))
    fq = str(tmp_path / "r.fq")
    S.write_fastq(fq, sim, names=["r%d" % i for i in range(n_reads)])
    res = ix.map_fastq(opt, fq, str(tmp_path / "o.sam"), want_hits=n_reads)
    n_hit = n_gapped = n_nohit = n_conv = 0
    for i, read in enumerate(reads):
        best, places, gapped = brute_best(T, read, opt)
        na, alns = ix.aln_one(opt, read, cap=512)
        h = res["hits"][i]
        if best is None:
            assert na == 0 and h.type == 0, (i, truth[i], alns[:1])
            n_nohit += 1
            continue
        assert na > 0, (i, truth[i], best)
        assert alns[0]["score"] == best, (i, truth[i], alns[0], best)
        assert alns[0]["units"] <= _costs(opt)["budget"]
        top = [a for a in alns[:512] if a["score"] == best]
        got = set()
        for a in top:
            got |= _aln_places(ix, a, n)
        assert got <= places, (i, truth[i], sorted(got - places)[:3])
        if not gapped:
            assert got == places, (i, truth[i], sorted(places - got)[:3])
            assert h.c1 == len(places)
        # the SAM-level record: one of the optimal placements, NM and CIGAR consistent with it
        assert h.type in (1, 2)
        cig = [(int(c) >> 4, int(c) & 0xf) for c in h.cigar[:h.n_cigar]] or [(L, 0)]
        span = sum(ln for ln, op in cig if op in (0, 2))
        if h.n_gapo == 0:
            assert (int(h.pos), int(h.strand), L) in places, (i, truth[i], h.pos, h.strand)
            assert h.score == best
        else:
            assert any(abs(int(h.pos) - p) <= 3 and int(h.strand) == s for p, s, _ in places), (i, truth[i], h.pos, h.strand, sorted(places)[:3])
            n_gapped += 1
        assert _cigar_ok(cig, read, T, int(h.pos), int(h.strand)) == h.nm, (i, truth[i], cig, h.nm)
        assert span == L + sum(ln for ln, op in cig if op == 2) - sum(ln for ln, op in cig if op == 1)
        n_hit += 1
        if mode == "profile" and best in (3, 6, 9, 12, 15):
            n_conv += 1
    # the sample must actually exercise what this test is for
    assert n_hit >= 60 and n_gapped >= 8, (n_hit, n_gapped, n_nohit)
    if mode == "profile":
        assert n_conv >= 10, n_conv
    print("%s: %d reads with a hit checked (%d of them gapped), %d without any alignment inside the budget (oracle agrees)" % (mode, n_hit, n_gapped, n_nohit))
