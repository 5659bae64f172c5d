"""order_probe.py -- how much of the search kernel's time is its tail?  (measurement aid, not a product path)

Runs the bench batch once with the counting kernel (per-read iteration counts), then times the search kernel on the same
reads handed out (a) in the product's order (by leading bases) and in input order, (b) longest search first by the TRUE iteration counts (an oracle order: the bound on
what any predictor of a read's cost could gain), (c) longest last (the worst case).
usage: python tools/order_probe.py [genome_mbp] [reads]"""
import os
import sys

import numpy as np

os.environ["PS_READ_ITERS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "para-suite_amd"))
import torch   # noqa: E402
import bench   # noqa: E402
import capi    # noqa: E402

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 3100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dev = torch.device("cuda", 0)
contigs = bench.gen_genome(torch, dev, mbp * 1_000_000, 24, 0x5EED0002)
fa = "/tmp/g_order.fa"
bench.write_fasta(fa, contigs)
torch.cuda.empty_cache()
ctx = capi.Ctx.build(fa)
rd = bench.gen_reads(torch, dev, contigs, n, 50, 0x5EED0003)
del contigs
torch.cuda.empty_cache()
P = np.array(bench.PROFILE); P[3, 1], P[3, 3] = 0.12, 0.87
ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1)
ctx.set_stats(True)
b = ctx.batch_from_codes(rd)
b.search()
it = ctx.read_iters().reshape(-1, capi.Ctx.RI_WORDS).astype(np.int64)[:, 0]
print("iterations per read: mean %.0f, percentiles 50/90/99/99.9/max %s" % (it.mean(), np.percentile(it, [50, 90, 99, 99.9, 100]).astype(int).tolist()), flush=True)
del b
ctx.set_stats(False)


def timed(order, label):
    codes = rd if order is None else rd[torch.from_numpy(order).to(rd.device)]
    bb = ctx.batch_from_codes(codes)
    ms = []
    for _ in range(3):
        bb.search()
        ms.append(bb.timing()["ms_backtrack"])
    print("%-34s ms_backtrack %s" % (label, [round(x, 1) for x in ms]), flush=True)
    del bb


timed(None, "leading-base order (the product's)")
os.environ["PS_KEEP_ORDER"] = "1"          # from here on the library keeps the order it is given
timed(None, "input order")
o = np.argsort(-it, kind="stable")
timed(o, "longest first (true counts)")
timed(o[::-1].copy(), "longest last")
# a coarse order: only the heaviest 1% / 5% moved to the front, the rest in input order
for frac in (0.01, 0.05):
    k = int(n * frac)
    heavy = np.zeros(n, dtype=bool); heavy[o[:k]] = True
    timed(np.concatenate([o[:k], np.flatnonzero(~heavy)]), "heaviest %.0f%% first" % (100 * frac))
