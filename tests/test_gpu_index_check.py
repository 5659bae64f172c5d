"""Exhaustive check of the product's FM index against the TEXT (VERDICT r2 #4b: the sampled checks of tests/index_props.py look at
~1e5 of up to 6.2e9 rows; at hg19 size the oracle adopts the product's BWT, so nothing else vouches for it).

ps_ctx_index_check (ps_kernels.hip: k_index_check) walks the whole LF cycle of the index, cut into arcs at the sampled rows: along
every arc each BWT symbol must be the text symbol in front of the current suffix -- the text being the packed forward strand (+ its
reverse complement), which the tests compare with the FASTA -- and every arc must arrive at a sample that holds the position counted
down to.  rows visited == n + 1 with no mismatch <=> the last column is the BWT of this text (a string is a BWT iff its LF map is one
cycle that spells the text), Occ / L2 are consistent with it and every SA sample is right.  Negative controls: one flipped BWT
symbol, one wrong SA sample, one wrong text base, each through a second context built from modified copies of the blobs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pac_matches_fasta(ctx, genome):
    """the packed text the check reads IS the FASTA (non-ACGT positions hold a random base: upstream's rule)"""
    import simulate as S
    info = ctx.info()
    pac = ctx.fetch(2)
    fwd = np.empty(pac.size * 4, dtype=np.uint8)
    for k in range(4):
        fwd[k::4] = (pac >> (6 - 2 * k)) & 3
    at = 0
    for _, asc in genome:
        codes = S.contig_codes(np.asarray(asc))
        keep = codes < 4
        assert np.array_equal(fwd[at:at + codes.size][keep], codes[keep])
        at += codes.size
    assert at == info.l_pac


@pytest.mark.parametrize("which", ["example", "multi", "mid"])
def test_every_row_against_the_text(which, request):
    import capi
    fx = request.getfixturevalue(which)
    ctx = capi.Ctx.build(fx["fa"])
    _pac_matches_fasta(ctx, fx["genome"])
    info = ctx.info()
    r = ctx.index_check()
    assert r["rows"] == info.seq_len + 1, r
    assert r["bad_symbols"] == 0 and r["bad_samples"] == 0, r
    assert 1 <= r["longest_arc"] < 2000, r


def test_the_check_sees_a_damaged_index(multi):
    """one symbol of the BWT, one SA sample, one base of the text: each must show"""
    import torch
    import capi
    ctx = capi.Ctx.build(multi["fa"])
    info = ctx.info()
    assert ctx.index_check()["bad_symbols"] == 0
    meta = ctx.meta()
    blobs = [ctx.fetch(i) for i in range(3)]             # 0: Occ blocks, 1: sampled SA, 2: packed text

    def check(mod):
        hb = [b.copy() for b in blobs]
        mod(hb)
        dev = [torch.from_numpy(b).cuda() for b in hb]
        c2 = capi.Ctx.from_blobs(meta, 0, [d.data_ptr() for d in dev], keep=dev)
        r = c2.index_check()
        c2.close()
        return r

    def flip_bwt(hb):                                    # a symbol's low plane bit in block 5 (the running counts stay: Occ no longer matches)
        w = hb[0].view("<u4").reshape(-1, 16)
        w[5, 4] ^= np.uint32(1 << 7)
    r = check(flip_bwt)
    assert r["bad_symbols"] > 0 or r["bad_samples"] > 0 or r["rows"] != info.seq_len + 1, r

    def wrong_sample(hb):
        w = hb[1].view("<u4")
        w[17] += np.uint32(1)
    r = check(wrong_sample)
    assert r["bad_samples"] > 0, r

    def wrong_base(hb):
        hb[2][100] ^= np.uint8(0x30)
    r = check(wrong_base)
    assert r["bad_symbols"] > 0, r
