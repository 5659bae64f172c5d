"""worker of tests/test_dist_gloo.py: one rank of a world_size-N CPU rehearsal of the sharded job.
The per-rank engine here is the oracle (CPU); the sharding and the stream-position chain are the
product's (para-suite_amd/sharding.py), the same functions bench.py drives on the GPUs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "para-suite_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist
    import orc
    import sharding
    work, n_reads = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    a, b = sharding.shard_range(n_reads, rank, world)
    # index "broadcast": rank 0 owns the FASTA bytes, every rank receives them and indexes its copy
    blob = [open(os.path.join(work, "g.fa"), "rb").read()] if rank == 0 else [None]
    dist.broadcast_object_list(blob, src=0)
    fa = os.path.join(work, "g.rank%d.fa" % rank)
    open(fa, "wb").write(blob[0])
    ix = orc.Index.from_fasta(fa)
    lines = open(os.path.join(work, "r.fq")).read().split("\n")
    fq = os.path.join(work, "r.rank%d.fq" % rank)
    open(fq, "w").write("\n".join(lines[4 * a:4 * b]) + "\n")
    out = os.path.join(work, "out.rank%d.sam" % rank)
    buf = torch.zeros(1, dtype=torch.int64)
    res = {}

    def advance(before):
        r = ix.map_fastq(orc.stock_opt("0.04"), fq, out, draws_before=before)
        res["after"] = r["draws_after"]
        return r["draws_after"]

    before, after = sharding.chain_stream_position(dist, rank, world, buf, advance)
    gathered = [None] * world
    dist.all_gather_object(gathered, (a, b, before, after))
    if rank == 0:
        open(os.path.join(work, "chain.txt"), "w").write(repr(gathered))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
