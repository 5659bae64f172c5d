"""backfill_probe.py -- can the next search launch fill the drain of the previous one?  (measurement aid)

Two batches of the bench workload on two lanes of work (two streams, two workspaces).  (a) one after the other; (b) both
launched at once from two threads (what `bench.py --sub-batches 2` does: the kernels share the CUs from the start);
(c) the second launched DELAY ms after the first: the first kernel's grid is resident before the second is submitted, so the
second's workgroups can only start where the first's retire.
usage: python tools/backfill_probe.py [genome_mbp] [reads_per_batch] [delay_ms ...]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "para-suite_amd"))
import torch   # noqa: E402
import bench   # noqa: E402
import capi    # noqa: E402

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 3100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
delays = [float(x) for x in sys.argv[3:]] or [0.0, 20.0, 100.0]
dev = torch.device("cuda", 0)
contigs = bench.gen_genome(torch, dev, mbp * 1_000_000, 24, 0x5EED0002)
fa = "/tmp/g_backfill.fa"
bench.write_fasta(fa, contigs)
torch.cuda.empty_cache()
ctx = capi.Ctx.build(fa)
rd = bench.gen_reads(torch, dev, contigs, 2 * n, 50, 0x5EED0003)
del contigs
torch.cuda.empty_cache()
P = np.array(bench.PROFILE); P[3, 1], P[3, 3] = 0.12, 0.87
ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1)
ctx.set_lanes(2)
A = ctx.batch_from_codes(rd[:n]); B = ctx.batch_from_codes(rd[n:])
A.search(); B.search()                     # warm-up: workspaces allocated


def span(*bs):
    t = [b.timing() for b in bs]
    return max(x["bt_end_ms"] for x in t) - min(x["bt_begin_ms"] for x in t)


for rep in range(2):
    t0 = time.perf_counter(); A.search(); B.search(); w = 1e3 * (time.perf_counter() - t0)
    print("one after the other: wall %.0f ms, kernels %.0f + %.0f ms" % (w, A.timing()["ms_backtrack"], B.timing()["ms_backtrack"]), flush=True)
for d in delays:
    for rep in range(2):
        def second():
            time.sleep(d * 1e-3)
            B.search()
        th = threading.Thread(target=second)
        t0 = time.perf_counter(); th.start(); A.search(); th.join(); w = 1e3 * (time.perf_counter() - t0)
        print("second launched %.0f ms after the first: wall %.0f ms, first launch start to last launch end %.0f ms (kernels %.0f / %.0f ms)" %
              (d, w, span(A, B), A.timing()["ms_backtrack"], B.timing()["ms_backtrack"]), flush=True)
