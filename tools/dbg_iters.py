import sys, os, time, numpy as np
os.environ['PS_READ_ITERS'] = '1'
sys.path.insert(0, 'para-suite_amd'); sys.path.insert(0, '.')
import capi, torch, bench
dev = torch.device('cuda', 0)
contigs = bench.gen_genome(torch, dev, 200_000_000, 8, 0x5EED0002)
fa = '/tmp/g200.fa'
bench.write_fasta(fa, contigs)
ctx = capi.Ctx.build(fa)
rd = bench.gen_reads(torch, dev, contigs, 2_000_000, 50, 0x5EED0003, indels=True)
for mode in ('stock', 'profile'):
    if mode == 'stock': ctx.set_stock('0.04')
    else:
        P = np.array(bench.PROFILE); P[3,1], P[3,3] = 0.12, 0.87
        ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1)
    for n in (2_000_000, 500_000, 100_000):
        b = ctx.batch_from_codes(rd[:n]); b.search(); tm = b.timing()
        it = ctx.read_iters().astype(np.int64)
        na = b.n_aln()
        q = np.percentile(it, [50, 90, 99, 99.9, 99.99, 100]).astype(int)
        print(mode, n, 'bt ms %.1f' % tm['ms_backtrack'], 'launches', tm['n_backtrack_launches'], 'iters mean %.0f' % it.mean(), 'pct', q.tolist(),
              'share of iters in top 1%%: %.2f' % (np.sort(it)[-n // 100:].sum() / it.sum()), 'unmapped mean iters %.0f' % it[na == 0].mean(), 'mapped %.0f' % it[na > 0].mean())
        b.free()
