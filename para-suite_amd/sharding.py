"""Multi-GPU sharding of a mapping job (one process per GPU).

Reads are independent, so ranks take contiguous ranges of the input and never exchange read data
(SURVEY.md §8e).  The one thing that is sequential in the reference's aligner is the tie-break RNG
of `bwa samse`: ONE drand48 stream over all reads in input order.  To keep the SAM identical for any
number of ranks, each rank hands the stream position (an integer: draws consumed so far) to the next
rank; only the reads whose draw count is data dependent sit on that chain (ps_batch_select_hard).
"""


def shard_range(n_items, rank, world):
    """contiguous range [a, b) of rank in a split of n_items into world shards of ceil(n/world)"""
    per = -(-n_items // world)
    a = min(n_items, rank * per)
    return a, min(n_items, a + per)


def chain_stream_position(dist, rank, world, buf, advance):
    """advance(draws_before) -> draws_after runs in rank order; buf is a 1-element int64 tensor
    (on the GPU for the nccl backend, on the CPU for gloo).  Returns (before, after)."""
    before = 0
    if world > 1 and rank > 0:
        dist.recv(buf, src=rank - 1)
        before = int(buf.item())
    after = int(advance(before))
    if world > 1 and rank < world - 1:
        buf.fill_(after)
        dist.send(buf, dst=rank + 1)
    return before, after
