"""per-read iteration counts of the backtracking kernel at bench scale (profiling aid; PS_READ_ITERS=1)"""
import sys, os, time, numpy as np
os.environ['PS_READ_ITERS'] = '1'
sys.path.insert(0, 'para-suite_amd'); sys.path.insert(0, '.')
import capi, torch, bench
mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dev = torch.device('cuda', 0)
contigs = bench.gen_genome(torch, dev, mbp * 1_000_000, 8, 0x5EED0002)
fa = '/tmp/g.fa'
bench.write_fasta(fa, contigs)
ctx = capi.Ctx.build(fa)
rd = bench.gen_reads(torch, dev, contigs, n, 50, 0x5EED0003, indels=True)
for mode in ('stock', 'profile'):
    if mode == 'stock': ctx.set_stock('0.04')
    else:
        P = np.array(bench.PROFILE); P[3,1], P[3,3] = 0.12, 0.87
        ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1)
    b = ctx.batch_from_codes(rd); b.search(); tm = b.timing()
    it = ctx.read_iters().reshape(-1, capi.Ctx.RI_WORDS).astype(np.int64)[:, 0]
    q = np.percentile(it, [50, 90, 99, 99.9, 99.99, 99.999, 100]).astype(int)
    srt = np.sort(it)
    print(mode, n, 'bt ms %.1f' % tm['ms_backtrack'], 'iters mean %.0f' % it.mean(), 'pct 50/90/99/99.9/99.99/99.999/max', q.tolist(),
          'share of iterations in the top 0.1%%: %.3f, top 1%%: %.3f' % (srt[-n // 1000:].sum() / it.sum(), srt[-n // 100:].sum() / it.sum()), flush=True)
    b.free()
    if mode == 'profile':
        na = None
        key = np.zeros(n, dtype=np.uint64)
        for j in range(16): key = (key << np.uint64(2)) | (rd[:, j] & 3).astype(np.uint64)
        order = np.argsort(key, kind='stable')          # the library orders a bin by the leading bases; read_iters is in that order
        rs = rd[order]
        cnt = [(rs == c).sum(1) for c in range(4)]
        for c in range(4):
            print('corr(iters, #%s) = %.3f' % ('ACGT'[c], np.corrcoef(it, cnt[c])[0, 1]))
        heavy = it > np.percentile(it, 99.9)
        print('heavy reads (top 0.1%%): mean #A %.1f #C %.1f #G %.1f #T %.1f ; all reads: %.1f %.1f %.1f %.1f' % tuple([cnt[c][heavy].mean() for c in range(4)] + [cnt[c].mean() for c in range(4)]))
        # a candidate predictor: C's and G's are the read bases reachable by the cheap substitutions (T->C on either strand)
        score = cnt[1] + cnt[2]
        for thr in (30, 32, 34, 36):
            sel = score >= thr
            print('score>=%d: %.3f%% of reads, hold %.1f%% of the top-0.1%% heavy reads, %.1f%% of all iterations' % (thr, 100 * sel.mean(), 100 * (sel & heavy).sum() / heavy.sum(), 100 * it[sel].sum() / it.sum()))
