"""GPU tier: the N>1 path of bench.py rehearsed with two ranks sharing the one GPU of the test box
(gloo backend: the index blobs are staged through host memory instead of RCCL/xGMI; everything else --
from_blobs contexts, read sharding, the chained tie-break stream -- is the code the 8-GPU run uses)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["sub_batches", "pipeline", "strong"])
def test_two_ranks_bench_rehearsal(tmp_path, mode):
    """two bench.py ranks (gloo, both on GPU 0) with the tie-break stream chained through the ranks -- "sub_batches": each rank maps
    its shard as two overlapping sub-batches (the chain also runs through them); "pipeline": the default, consecutive steps
    alternate between two batches in flight and take the chain in step order; "strong": BASELINE configs[3]'s shape -- ONE job of
    --reads reads, rank r maps the contiguous range sharding.shard_range gives it (the default of bench.py for N > 1; the
    other two modes are --scaling weak: --reads per rank).  The per-read records of both ranks, concatenated, equal those of ONE
    process mapping the same reads as one batch.  (RCCL / xGMI itself cannot run on the
    one-GPU test box: no N > 1 hardware number exists.)"""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench as B
    import capi
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dump = str(tmp_path / "hits")
    mbp, n_reads = 8, 60000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", {"sub_batches": "29655", "pipeline": "29657", "strong": "29659"}[mode], os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
           "--genome-mbp", str(mbp), "--contigs", "4", "--reads", str(n_reads if mode != "strong" else 2 * n_reads + 1), "--cpu-sample", "0", "--dump-hits", dump]
    cmd += ["--scaling", "strong" if mode == "strong" else "weak"]
    cmd += ["--steps", "1", "--warmup", "1", "--sub-batches", "2"] if mode == "sub_batches" else ["--steps", "4", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.split("\n") if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if mode == "strong" else "weak") and d["value"] > 0
    total = 2 * n_reads + 1 if mode == "strong" else 2 * n_reads           # strong: an odd total, so the ranges differ in size
    assert d["config"]["reads_total"] == total and d["config"]["reads_per_gpu"] == -(-total // 2 if mode == "strong" else -n_reads)
    assert abs(d["value"] - total * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]      # whole-job reads over the slowest rank's time
    assert (d["config"]["sub_batches"], d["config"]["pipeline"]) == ((2, 1) if mode == "sub_batches" else (1, 2))
    assert d["mapped_frac"] > 0.8
    got = np.concatenate([np.load(os.path.join(dump, "hits_rank%d.npy" % k)) for k in range(2)])
    # the same genome and the same reads (rank r draws its reads from seed 0x5EED0003 + r), one process, one batch
    dev = torch.device("cuda", 0)
    contigs = B.gen_genome(torch, dev, mbp * 1_000_000, 4, 0x5EED0002)     # 2 Mbp contigs: with the two-copy segments
    fa = str(tmp_path / "g.fa")
    B.write_fasta(fa, contigs)
    ctx = capi.Ctx.build(fa, device=0)
    P = np.array(B.PROFILE); P[3, 1], P[3, 3] = 0.12, 0.87
    ctx.set_profile(P, B.INS_RATE, B.DEL_RATE, -1)
    if mode == "strong":
        codes = B.gen_reads(torch, dev, contigs, total, 50, 0x5EED0003)
    else:
        codes = np.concatenate([B.gen_reads(torch, dev, contigs, n_reads, 50, 0x5EED0003 + k) for k in range(2)])
    one = ctx.batch_from_codes(codes)
    one.run(8)
    exp = one.hits()
    assert len(got) == len(exp) == total
    for f in ("pos", "sa", "type", "strand", "mapq", "n_mm", "n_gapo", "c1", "c2", "n_cigar", "n_multi"):
        assert np.array_equal(got[f], exp[f]), f
    assert (exp["c1"] > 1).sum() > 100       # reads that did need the random tie-break, on both sides of every hand-over


def test_sharded_batches_equal_single_batch(multi, workdir):
    """two half batches with the stream position chained == one batch (what the ranks do, in one process)"""
    import numpy as np
    import capi
    import simulate as S
    sim = S.simulate_reads(multi["genome"], 3000, 50, seed=55, indel_scale=30)
    codes = sim["codes"]
    ctx = capi.Ctx.build(multi["fa"])
    ctx.set_stock("0.04")
    whole = ctx.batch_from_codes(codes)
    whole.run(4)
    hw = whole.hits()
    parts = []
    before = 0
    for a, b in ((0, 1300), (1300, 3000)):
        sb = ctx.batch_from_codes(codes[a:b])
        sb.search()
        before = sb.select_hard(before)
        sb.select_easy(4)
        sb.locate()
        parts.append(sb.hits())
    hp = np.concatenate(parts)
    for f in ("pos", "sa", "type", "strand", "mapq", "n_mm", "n_gapo", "c1", "c2", "n_cigar", "n_multi"):
        assert np.array_equal(hw[f], hp[f]), f
    assert (hw["c1"] > 1).sum() > 0          # some reads did need the random tie-break
