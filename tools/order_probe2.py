"""order_probe2.py -- what is the search launch's tail worth, and what predicts a read's effort?  (measurement aid, not a product path)

tools/order_probe.py (round 2) applied the per-read iteration counts -- which the library returns in ITS order of the reads
(a bin's leading-base order) -- as if they were in input order, so its "longest first" orders were random ones.  This probe maps
the counts back to input order first (the library's sort is a stable sort on the leading 16 bases: reproduced here), then times
the search kernel on the same reads handed out
  product     leading-base order (what the library does)
  oracle-K    K effort classes by the TRUE iteration counts, heaviest class first, leading-base order inside a class
  oracle-sort fully sorted by true count, longest first (no locality left)
  pred-*      classes by what is known BEFORE the search (the width stage's lower bounds, base composition), heaviest first
and writes a sample of (iterations, bounds, outcome, base counts) per read for fitting a predictor offline.
usage: python tools/order_probe2.py [genome_mbp] [reads] [out_prefix]"""
import os
import sys

import numpy as np

os.environ["PS_READ_ITERS"] = "1"
os.environ["PS_ORDER"] = "1"            # the counting pass records the estimate the library orders by
for kv in sys.argv[4:]:                  # further NAME=value settings for the library (e.g. PS_ORDER_WPIN=4)
    os.environ[kv.split("=")[0]] = kv.split("=")[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "para-suite_amd"))
import torch   # noqa: E402
import bench   # noqa: E402
import capi    # noqa: E402

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 3100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
outp = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/order2"
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

# the library's order of a bin: stable sort on the leading 16 bases (code & 3)
rd_h = np.asarray(rd)
key = np.zeros(n, dtype=np.uint64)
for j in range(16):
    key = (key << np.uint64(2)) | (rd_h[:, j] & 3).astype(np.uint64)
lib_order = np.argsort(key, kind="stable")          # lib_order[p] = input index of the read at sorted position p

ctx.set_stats(True)
b = ctx.batch_from_codes(rd)
b.search()
ri = ctx.read_iters().reshape(-1, capi.Ctx.RI_WORDS)
assert ri.shape[0] == n
prof = np.empty_like(ri)
prof[lib_order] = ri                                # now in input order
it = prof[:, 0].astype(np.int64)
estA = ((prof[:, 2] >> 16) & 0x7f).astype(np.int64); estB = ((prof[:, 2] >> 24) & 0x7f).astype(np.int64); rsA = ((prof[:, 2] >> 23) & 1).astype(bool); rsB = ((prof[:, 2] >> 31) & 1).astype(bool)
est = np.where(~rsA, np.where(~rsB & (estB < estA), estB, estA), np.where(~rsB, estB, np.maximum(estA, estB)))     # the library's rule (ps_effort.hip)
d_read = (prof[:, 2] & 0x7f).astype(np.int64); d_seed = ((prof[:, 2] >> 8) & 0x7f).astype(np.int64); 
best = (prof[:, 3] & 0xff).astype(np.int64); budget1 = ((prof[:, 3] >> 8) & 0xff).astype(np.int64); n_hits = (prof[:, 3] >> 16).astype(np.int64)
print("iterations per read: mean %.0f, percentiles 50/90/99/99.9/max %s" % (it.mean(), np.percentile(it, [50, 90, 99, 99.9, 100]).astype(int).tolist()), flush=True)
del b
ctx.set_stats(False)
cnt = [(rd_h == c).sum(1) for c in range(4)]
cg = cnt[1] + cnt[2]


def table(name, v):
    print("by %s: value: share of reads, mean iterations, share of all iterations, P99 iterations" % name)
    for x in np.unique(v):
        s = v == x
        if s.sum() < 100: continue
        print("   %4d: %6.2f%%  %8.0f  %6.2f%%  %8d" % (x, 100 * s.mean(), it[s].mean(), 100 * it[s].sum() / it.sum(), np.percentile(it[s], 99)))


table("estimated best score (k_effort)", est)
table("D bound of the read", d_read)
table("D bound of the seed", d_seed)
table("best score found (255: none)", best)
table("final budget", budget1)
table("#C + #G (coarse: //4)", cg // 4)
for nm, f in (("est", est), ("min(est,16)", np.minimum(est, 16)), ("d_read", d_read), ("d_seed", d_seed), ("best", np.where(best == 255, 40, best)), ("budget1", budget1), ("cg", cg), ("log it", np.log1p(it))):
    print("corr(log iterations, %s) = %.3f" % (nm, np.corrcoef(np.log1p(it), f)[0, 1]))
ns = min(n, 2_000_000)
np.savez_compressed(outp + "_sample.npz", it=it[:ns].astype(np.uint32), d_read=d_read[:ns].astype(np.uint8), d_seed=d_seed[:ns].astype(np.uint8), best=best[:ns].astype(np.uint8),
                    budget1=budget1[:ns].astype(np.uint8), estA=estA[:ns].astype(np.uint8), estB=estB[:ns].astype(np.uint8), rsA=rsA[:ns], rsB=rsB[:ns], cw=np.ascontiguousarray(prof[:ns, 4:17]).view(np.uint8).reshape(ns, 52), n_hits=n_hits[:ns].astype(np.uint16), reads=rd_h[:ns].astype(np.uint8))

print("confusion: rows = true final budget, columns = min(est + 8, 24) (share of reads, %)")
eb = np.minimum(est + 8, 24)
for bt in np.unique(budget1):
    row = [100.0 * ((budget1 == bt) & (eb == x)).sum() / n for x in np.unique(eb)]
    print("   %3d: " % bt + " ".join("%5.2f" % v for v in row))
print("   cols: " + " ".join("%5d" % x for x in np.unique(eb)))

rank_lib = np.empty(n, dtype=np.int64); rank_lib[lib_order] = np.arange(n)


def timed(order, label, reps=3, env=None):
    env = env or {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    codes = rd_h if order is None else rd_h[order]
    bb = ctx.batch_from_codes(codes)
    ms, mw = [], []
    for _ in range(reps):
        bb.search()
        ms.append(bb.timing()["ms_backtrack"]); mw.append(bb.timing()["ms_width"])
    print("%-58s ms_backtrack %s  ms_width+order %s" % (label, [round(x, 1) for x in ms], [round(x, 1) for x in mw]), flush=True)
    del bb
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v


def classes_then_lib(cls):                  # heaviest class (largest value) first, the library's order inside a class
    return np.lexsort((rank_lib, -cls))


# the library orders by leading bases itself (PS_KEEP_ORDER unset); PS_ORDER = its own effort order on top
timed(None, "product, PS_ORDER=0", env={"PS_ORDER": "0"})
timed(None, "PS_ORDER=1 (expected-nodes model, as set)", env={"PS_ORDER": "1"})
if len(sys.argv) > 4:
    raise SystemExit(0)
os.environ["PS_KEEP_ORDER"] = "1"          # from here on the library keeps the order it is given
os.environ["PS_ORDER"] = "0"
q = np.searchsorted(np.quantile(it, np.arange(1, 8) / 8), it, side="right")
timed(classes_then_lib(q), "oracle-8 (true counts, classes)")
timed(np.argsort(-it, kind="stable"), "oracle-sort (longest first, no locality)")
