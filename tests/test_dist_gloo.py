"""CPU tier, world_size 2 (and 3) over gloo: reads sharded in contiguous ranges, the tie-break stream
position chained through the ranks, SAM identical to the single-process run."""
import ast
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range():
    import sharding
    assert [sharding.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [sharding.shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert sharding.shard_range(0, 0, 2) == (0, 0)
    cover = [sharding.shard_range(1001, r, 8) for r in range(8)]
    assert cover[0][0] == 0 and cover[-1][1] == 1001 and all(cover[i][1] == cover[i + 1][0] for i in range(7))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_job_matches_single_process(example, tmp_path, world):
    import orc
    import simulate as S
    work = str(tmp_path)
    S.write_fasta(os.path.join(work, "g.fa"), example["genome"])
    n = 900
    sim = S.simulate_reads(example["genome"], n, 50, seed=77, indel_scale=20)
    S.write_fastq(os.path.join(work, "r.fq"), sim)
    whole = os.path.join(work, "whole.sam")
    r = example["orc_index"].map_fastq(orc.stock_opt("0.04"), os.path.join(work, "r.fq"), whole)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29611 + world), os.path.join(HERE, "dist_worker.py"), work, str(n)]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    got = []
    for k in range(world):
        got += [l for l in open(os.path.join(work, "out.rank%d.sam" % k)) if not l.startswith("@")]
    assert got == [l for l in open(whole) if not l.startswith("@")]
    chain = ast.literal_eval(open(os.path.join(work, "chain.txt")).read())
    assert chain[0][2] == 0 and chain[-1][3] == r["draws_after"]
    assert all(chain[i][3] == chain[i + 1][2] for i in range(world - 1))          # each rank starts where its predecessor stopped
