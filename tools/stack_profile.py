"""per-read stack footprint of the search kernel at bench scale (profiling aid: the counting kernel's per-read profile,
PS_READ_ITERS=1): how many stack slots a read uses, which sizes the private slices need.
usage: python tools/stack_profile.py [genome_mbp] [reads]"""
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
fa = "/tmp/g_stack.fa"
bench.write_fasta(fa, contigs)
torch.cuda.empty_cache()
ctx = capi.Ctx.build(fa)
rd = bench.gen_reads(torch, dev, contigs, n, 50, 0x5EED0003, indels=True)
del contigs
torch.cuda.empty_cache()
P = np.array(bench.PROFILE); P[3, 1], P[3, 3] = 0.12, 0.87
ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1)
b = ctx.batch_from_codes(rd)
b.search()
prof = ctx.read_iters().reshape(-1, capi.Ctx.RI_WORDS).astype(np.int64)
it, slots = prof[:, 0], prof[:, 1]
print("reads %d: iterations mean %.0f; stack slots used per read: mean %.0f, percentiles 50/90/99/99.9/99.99/max %s" %
      (len(it), it.mean(), slots.mean(), np.percentile(slots, [50, 90, 99, 99.9, 99.99, 100]).astype(int).tolist()))
for cap in (256, 512, 1024, 2048, 4096, 8192, 12288, 16384):
    over = slots + 9 > cap
    print("  private slice of %5d entries (%4d KB/lane, %5.1f GB for 262144 lanes): %.3f%% of the reads outgrow it, they hold %.1f%% of all iterations" %
          (cap, cap * 16 // 1024, cap * 16 * 262144 / 1e9, 100 * over.mean(), 100 * it[over].sum() / it.sum()))
