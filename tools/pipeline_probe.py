"""pipeline_probe.py -- steps of the bench workload with two batches in flight (measurement aid).

(a) K steps one after the other on one batch (what bench.py does by default: search, selection, locate);
(b) K steps alternating between two batches on two lanes of work, each driven by its own thread, the second thread started a
    little later: the search launch of step k+1 is submitted while step k's kernel still runs and takes the slots its retiring
    workgroups leave (tools/backfill_probe.py), step k's short later stages run inside that hand-over.
usage: python tools/pipeline_probe.py [genome_mbp] [reads] [steps]"""
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
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
contigs = bench.gen_genome(torch, dev, mbp * 1_000_000, 24, 0x5EED0002)
fa = "/tmp/g_pipe.fa"
bench.write_fasta(fa, contigs)
torch.cuda.empty_cache()
ctx = capi.Ctx.build(fa)
rd = bench.gen_reads(torch, dev, contigs, n, 50, 0x5EED0003)
del contigs
torch.cuda.empty_cache()
P = np.array(bench.PROFILE); P[3, 1], P[3, 3] = 0.12, 0.87
ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1)
ctx.set_lanes(2)
B = [ctx.batch_from_codes(rd), ctx.batch_from_codes(rd)]


def one(b):
    b.search(); b.select_hard(0); b.select_easy(16); b.locate()


for b in B:
    one(b)                                  # warm-up, workspaces allocated
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    one(B[0])
torch.cuda.synchronize()
seq = time.perf_counter() - t0
print("one batch, %d steps one after the other: %.0f ms per step" % (K, 1e3 * seq / K), flush=True)
h0 = B[0].hits().copy()

for stagger in (0.05, 0.3):
    def worker(j):
        if j:
            time.sleep(stagger)
        for k in range(j, K, 2):
            one(B[j])
    th = [threading.Thread(target=worker, args=(j,)) for j in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    pip = time.perf_counter() - t0
    print("two batches in flight (second thread %.0f ms later), %d steps: %.0f ms per step" % (1e3 * stagger, K, 1e3 * pip / K), flush=True)
assert (B[1].hits() == h0).all() and (B[0].hits() == h0).all()
print("hits identical")
