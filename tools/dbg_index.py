import sys, os, time, numpy as np
sys.path.insert(0, 'para-suite_amd'); sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import capi, simulate as S, orc, torch, bench
dev = torch.device('cuda', 0)
contigs = bench.gen_genome(torch, dev, 200_000_000, 8, 0x5EED0002)
fa = '/tmp/g200.fa'
bench.write_fasta(fa, contigs)
t = time.time(); ctx = capi.Ctx.build(fa); print('gpu build', time.time() - t, ctx.info().sa_rounds, flush=True)
codes = torch.cat([c for _, c in contigs]).cpu().numpy()
rng = np.random.default_rng(1)
st = rng.integers(0, codes.size - 60, 50000)
reads = codes[st[:, None] + np.arange(50)[None, :]]
ok = (reads < 4).all(1)
reads = reads[ok]; st = st[ok]
ctx.set_stock('0')
b = ctx.batch_from_codes(reads); b.run(4); h = b.hits()
print('exact reads mapped', (h['type'] != 0).mean(), 'pos ok', (h['pos'] == st).mean(), flush=True)
rd = bench.gen_reads(torch, dev, contigs, 100000, 50, 7, indels=True)
for mode in ('stock', 'profile'):
    if mode == 'stock': ctx.set_stock('0.04')
    else:
        P = np.array(bench.PROFILE); P[3,1], P[3,3] = 0.12, 0.87
        ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1)
    b = ctx.batch_from_codes(rd); b.run(4); h = b.hits()
    print(mode, 'mapped', (h['type'] != 0).mean(), 'nm hist', np.bincount(h['n_mm'][h['type'] != 0])[:6], b.timing())
    na = b.n_aln(); print('n_aln hist', np.bincount(np.minimum(na, 10)))
# hamming check of a few reads against the genome around expected position is not available (gen_reads hides truth);
# instead count mismatches of exact-window reads with 2 forced substitutions
rr = reads[:20000].copy(); rr[:, 10] = (rr[:, 10] + 1) & 3; rr[:, 30] = (rr[:, 30] + 2) & 3
ctx.set_stock('0.04')
b = ctx.batch_from_codes(rr); b.run(4); h = b.hits()
print('2-mismatch reads mapped', (h['type'] != 0).mean(), 'pos ok', (h['pos'] == st[:20000]).mean())
