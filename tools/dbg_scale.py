import sys, os, time, numpy as np
sys.path.insert(0, 'para-suite_amd'); sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import capi, torch, bench
dev = torch.device('cuda', 0)
contigs = bench.gen_genome(torch, dev, 200_000_000, 8, 0x5EED0002)
fa = '/tmp/g200.fa'
bench.write_fasta(fa, contigs)
ctx = capi.Ctx.build(fa)
rd = bench.gen_reads(torch, dev, contigs, 2_000_000, 50, 0x5EED0003, indels=True)
print('N frac per 100k block', np.round((rd == 4).reshape(20, -1).mean(1), 4), 'max code', rd.max())
ctx.set_stock('0.04')
small = ctx.batch_from_codes(rd[1_000_000:1_100_000]); small.run(4); hs = small.hits()
print('small@1M mapped', (hs['type'] != 0).mean())
for blocks in (0, 256, 2048):
    ctx.set_tiers(None, None, blocks)
    big = ctx.batch_from_codes(rd); big.search(); nab = big.n_aln()
    print('bt_blocks', blocks, 'n_aln>0 per 100k block', np.round((nab > 0).reshape(20, -1).mean(1), 3), big.timing()['ms_backtrack'])
    big.free()
