#!/usr/bin/env python3
"""bench.py -- reads/s of the mapping hot path on MI355X.

One "step" = one pass of the hot path (width kernel -> backtracking kernel -> tie-break selection ->
SA-walk kernel -> banded-DP kernel) over one batch of synthetic PAR-CLIP reads that is already
packed and resident in HBM.  Workload at N=1: BASELINE.json configs[2] -- 10 M x 50 bp simulated
PAR-CLIP reads, full difference-tolerant search + gapped extension -- against a synthetic genome of
hg19's size (--genome-mbp, default 3100 in 24 contigs: 6.2e9 BWT rows, which is why rows are 33-bit;
hg19 itself is not on the box and there is no network to fetch it).
N>1: one process per GPU, the FM index built on rank 0 and broadcast once with RCCL.  --scaling strong (the
default, BASELINE.json configs[3]): the --reads reads are ONE job, rank r maps the contiguous range
ceil(reads/N)*r .. (sharding.shard_range); --scaling weak: every rank maps its own --reads reads.  No
data-path collective either way; the only exchange is one integer per rank that chains the tie-break RNG
stream in input order.

Launch:  python bench.py [--gpus 1] [--steps K] [--warmup W]
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "para-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

AT = 0.295
PROFILE = [[0.990, 0.004, 0.003, 0.003], [0.004, 0.990, 0.003, 0.003], [0.006, 0.010, 0.977, 0.007],
           [0.005, 0.005, 0.003, 0.987]]
SITE_FREQ = [0.66, 0.24, 0.08, 0.04]
INS_RATE, DEL_RATE = 2.1e-5, 5.9e-4


def gen_genome(torch, dev, total_bp, n_contigs, seed):
    """uint8 codes (0..3, 4 = N) per contig on the device; same stream on every rank."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    per = total_bp // n_contigs
    out = []
    for c in range(n_contigs):
        n = per if c < n_contigs - 1 else total_bp - per * (n_contigs - 1)
        codes = torch.empty(n, dtype=torch.uint8, device=dev)
        for a in range(0, n, 1 << 27):
            b = min(n, a + (1 << 27))
            u = torch.rand(b - a, generator=g, device=dev)
            codes[a:b] = ((u >= AT).to(torch.uint8) + (u >= 0.5).to(torch.uint8) + (u >= 1.0 - AT).to(torch.uint8))
        if n > 400000:
            # interspersed repeats: one 2 kb segment in every 50 kb is a copy (1 % diverged) of a segment elsewhere in
            # the contig, so ~4 % of the genome is two-copy sequence -- reads with several equally good hits, tied
            # suffixes for the index builder -- instead of an i.i.d. text in which every 28-mer is unique
            slot, seg = 50_000, 2_000
            R = n // slot - 1
            dst = torch.arange(R, device=dev, dtype=torch.int64) * slot + 20_000
            src = (torch.rand(R, generator=g, device=dev, dtype=torch.float64) * (n - seg - 1)).long()
            ar = torch.arange(seg, device=dev, dtype=torch.int64)[None, :]
            for a0 in range(0, R, 256):                 # small index tensors (large advanced-indexing calls misbehave on this stack)
                sl = slice(a0, min(R, a0 + 256))
                piece = codes[src[sl, None] + ar]
                mut = torch.rand(piece.shape, generator=g, device=dev) < 0.01
                piece = torch.where(mut, (piece + torch.randint(1, 4, piece.shape, generator=g, device=dev, dtype=torch.uint8)) & 3, piece)
                codes[dst[sl, None] + ar] = piece
            run = 20000
            mid = int(n * 0.4)
            codes[:run // 2] = 4
            codes[mid:mid + run] = 4
            codes[n - run // 2:] = 4
        out.append(("chr%d" % (c + 1), codes))
    return out


def add_repeats(torch, dev, contigs, seed, share=0.45, info=None, satellite_reps=400):
    """Repeat structure of a mammalian genome on top of gen_genome's text (bench.py --genome-profile repeats): interspersed
    families -- 300-bp units (SINE-like: one family with a copy per ~2.6 kb, minor ones with 10^3-10^5 copies at full scale)
    and 6-kb units (LINE-like, most copies truncated at their 5' end) -- every copy 2-20 % diverged from its family's consensus
    and on either strand; microsatellites (1-6 bp motifs, 10-60 units); a few satellite arrays (171-bp unit in tandem, 1-2 %
    diverged).  `share` of the bases end up inside interspersed copies (hg19: ~45 %).  Copy numbers scale with the genome so
    that a small test genome has the same density.  Deterministic in (seed, sizes); the N runs of gen_genome are restored.
    info (a dict, optional) receives where the satellite arrays went: info["satellites"] = [(contig index, start, length)]."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    total = sum(c.numel() for _, c in contigs)
    fam = []                                   # (unit length, share of the genome, divergence range, truncated copies)
    fam.append((300, 0.115, (0.05, 0.18), False))
    for k in range(6):
        fam.append((300, 0.012, (0.02 + 0.02 * k, 0.06 + 0.025 * k), False))
    for k in range(5):
        fam.append((6000, 0.045, (0.02 + 0.03 * k, 0.08 + 0.03 * k), True))
    scale = share / sum(f[1] for f in fam) / 0.80          # later copies overwrite earlier ones: ~20 % of the placed bases are lost
    cons = [torch.randint(0, 4, (u,), generator=g, device=dev, dtype=torch.uint8) for u, _, _, _ in fam]
    comp = torch.tensor([3, 2, 1, 0], dtype=torch.uint8, device=dev)
    for ci, (name, codes) in enumerate(contigs):
        n = codes.numel()
        if n < 20000:
            continue
        holes = codes == 4
        for (u, sh, (d0, d1), trunc), cs in zip(fam, cons):
            copies = int(n * sh * scale / (u * (0.55 if trunc else 1.0)))
            step = max(1, (1 << 22) // u)
            for a0 in range(0, copies, step):
                m = min(step, copies - a0)
                dst = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (n - u - 1)).long()
                div = d0 + (d1 - d0) * torch.rand(m, generator=g, device=dev)
                piece = cs[None, :].expand(m, u)
                mut = torch.rand((m, u), generator=g, device=dev) < div[:, None]
                piece = torch.where(mut, (piece + torch.randint(1, 4, (m, u), generator=g, device=dev, dtype=torch.uint8)) & 3, piece)
                rev = torch.rand(m, generator=g, device=dev) < 0.5
                piece = torch.where(rev[:, None], comp[piece.flip(1).long()], piece)
                ar = torch.arange(u, device=dev, dtype=torch.int64)[None, :]
                if trunc:                                    # keep the last `keep` bases of the unit (5'-truncated insertions)
                    keep = (u * (0.1 + 0.9 * torch.rand(m, generator=g, device=dev) ** 2)).long().clamp_(50, u)
                    ok = ar >= (u - keep)[:, None]
                    idx = (dst[:, None] + ar)[ok]
                    codes[idx] = piece[ok]
                else:
                    codes[(dst[:, None] + ar).reshape(-1)] = piece.reshape(-1)
        # microsatellites: one per ~15 kb
        k = max(1, n // 15000)
        for a0 in range(0, k, 4096):
            m = min(4096, k - a0)
            mot_len = torch.randint(1, 7, (m,), generator=g, device=dev)
            units = torch.randint(10, 61, (m,), generator=g, device=dev)
            motif = torch.randint(0, 4, (m, 6), generator=g, device=dev, dtype=torch.uint8)
            ar = torch.arange(360, device=dev, dtype=torch.int64)[None, :]
            seq = torch.gather(motif, 1, ar % mot_len[:, None])
            ok = ar < (mot_len * units)[:, None]
            dst = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (n - 361)).long()
            codes[(dst[:, None] + ar)[ok]] = seq[ok]
        # satellite arrays: ~0.3 % of the contig in arrays of 171-bp units
        n_arr = max(1, int(n * 0.003) // (171 * satellite_reps))
        for _ in range(n_arr):
            unit = torch.randint(0, 4, (171,), generator=g, device=dev, dtype=torch.uint8)
            reps = satellite_reps
            arr = unit.repeat(reps)
            mut = torch.rand(arr.numel(), generator=g, device=dev) < 0.015
            arr = torch.where(mut, (arr + torch.randint(1, 4, (arr.numel(),), generator=g, device=dev, dtype=torch.uint8)) & 3, arr)
            at = int(torch.rand(1, generator=g, device=dev).item() * (n - arr.numel() - 1))
            codes[at:at + arr.numel()] = arr
            if info is not None:
                info.setdefault("satellites", []).append((ci, at, int(arr.numel())))
        codes[holes] = 4
    return contigs


def write_fasta(path, contigs, width=50):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for name, codes in contigs:
            f.write(b">" + name.encode() + b"\n")
            n = codes.numel()
            for a in range(0, n, 50_000_000):
                b = min(n, a + 50_000_000)
                asc = lut[codes[a:b].cpu().numpy()]
                full = asc.size // width * width
                if full:
                    body = np.empty((full // width, width + 1), dtype=np.uint8)
                    body[:, :width] = asc[:full].reshape(-1, width)
                    body[:, width] = 10
                    f.write(body.tobytes())
                if full < asc.size:
                    f.write(asc[full:].tobytes() + b"\n")


def gen_reads(torch, dev, contigs, n_reads, L, seed, bound=0.6, indels=True):
    """PAR-CLIP reads with the distribution of para-suite_amd/simulate.py, generated on the device.
    Returns uint8 codes [n_reads, L] on the host."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sizes = torch.tensor([c.numel() for _, c in contigs], dtype=torch.float64)
    flat = torch.cat([c for _, c in contigs])
    offs = torch.cumsum(torch.tensor([0] + [c.numel() for _, c in contigs[:-1]]), 0).to(dev)
    span = L + 2
    out = torch.empty((n_reads, L), dtype=torch.uint8)
    comp = torch.tensor([3, 2, 1, 0, 4], dtype=torch.uint8, device=dev)
    cdf = torch.cumsum(torch.tensor(PROFILE, dtype=torch.float32, device=dev), 1)
    site = torch.tensor(SITE_FREQ, dtype=torch.float32, device=dev)
    ar = torch.arange(span, device=dev)
    jj = torch.arange(L, device=dev)[None, :]
    chunk = 500_000          # larger chunks trip a torch-ROCm indexing limit (rows beyond 655360 came back wrong)
    for a in range(0, n_reads, chunk):
        n = min(chunk, n_reads - a)
        cidx = torch.multinomial((sizes / sizes.sum()).float().to(dev), n, replacement=True, generator=g)
        csz = torch.tensor([c.numel() for _, c in contigs], device=dev)[cidx]
        start = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * (csz - span)).long()
        win = flat[(offs[cidx] + start)[:, None] + ar[None, :]]
        for _ in range(8):                                    # re-draw windows that touch an N run
            bad = (win == 4).any(1)
            nb = int(bad.sum())
            if nb == 0:
                break
            s2 = (torch.rand(nb, generator=g, device=dev, dtype=torch.float64) * (csz[bad] - span)).long()
            start[bad] = s2
            win[bad] = flat[(offs[cidx[bad]] + s2)[:, None] + ar[None, :]]
        win[win == 4] = 0
        strand = torch.rand(n, generator=g, device=dev) < 0.5
        isb = torch.rand(n, generator=g, device=dev) < bound
        has_ins = torch.zeros(n, dtype=torch.bool, device=dev)
        has_del = torch.zeros(n, dtype=torch.bool, device=dev)
        if indels:
            u = torch.rand(n, generator=g, device=dev)
            p_ins, p_del = 1 - (1 - INS_RATE) ** L, 1 - (1 - DEL_RATE) ** L
            has_ins = u < p_ins
            has_del = (~has_ins) & (u < p_ins + p_del)
        ipos = (6 + torch.rand(n, generator=g, device=dev) * max(L - 12, 1)).long()
        ref_len = L + has_del.long() - has_ins.long()
        src = torch.where(strand[:, None], ref_len[:, None] - 1 - ar[None, :], ar[None, :]).clamp_(0, span - 1)
        true = torch.gather(win, 1, src)
        true = torch.where(strand[:, None], comp[true.long()], true)
        tpos = (jj + (has_del[:, None] & (jj >= ipos[:, None])).long() - (has_ins[:, None] & (jj > ipos[:, None])).long()).clamp_(0, span - 1)
        read = torch.gather(true, 1, tpos)
        ins_here = has_ins[:, None] & (jj == ipos[:, None])
        read = torch.where(ins_here, torch.randint(0, 4, (n, L), generator=g, device=dev, dtype=torch.uint8), read)
        # T->C conversions on bound reads: up to 4 random T sites, site j converted with SITE_FREQ[j]
        prio = torch.where(read == 3, torch.rand((n, L), generator=g, device=dev), torch.full((n, L), 2.0, device=dev))
        val, order = torch.topk(prio, 4, dim=1, largest=False)
        conv = (val < 1.5) & (torch.rand((n, 4), generator=g, device=dev) < site[None, :]) & isb[:, None]
        read.scatter_(1, order, torch.where(conv, torch.ones_like(order, dtype=torch.uint8), torch.gather(read, 1, order)))
        # sequencing errors by the profile row of the (converted) base
        u = torch.rand((n, L), generator=g, device=dev)
        row = cdf[read.long()]
        read = ((u > row[..., 0]).to(torch.uint8) + (u > row[..., 1]).to(torch.uint8) + (u > row[..., 2]).to(torch.uint8))
        out[a:a + n] = read.cpu()
    return out.numpy()


def write_fastq_fast(path, codes, qual_char=73):
    """fixed-width records straight from the code matrix (names r00000000 ...): numpy, no per-read Python"""
    n, L = codes.shape
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    rec = np.empty((n, 1 + 9 + 1 + L + 3 + L + 1), dtype=np.uint8)
    idx = np.arange(n, dtype=np.int64)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    rec[:, 2:10] = ((idx[:, None] // (10 ** np.arange(7, -1, -1, dtype=np.int64))[None, :]) % 10 + 48).astype(np.uint8)
    rec[:, 10] = 10
    rec[:, 11:11 + L] = lut[codes]
    rec[:, 11 + L:14 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 14 + L:14 + 2 * L] = qual_char
    rec[:, 14 + 2 * L] = 10
    with open(path, "wb") as f:
        for a in range(0, n, 1 << 20):
            f.write(rec[a:a + (1 << 20)].tobytes())


def write_java_profile(prefix, P, ins, dele):
    """the two files ErrorProfiling.java:504-531,545-591 writes (Double.toString values + tab; ins TAB del, no newline)"""
    def jd(v):
        v = float(v)
        if v != v:
            return "NaN"
        if v != 0 and (abs(v) < 1e-3 or abs(v) >= 1e7):
            m, e = ("%.16e" % v).split("e")
            m = repr(float(m))
            return "%sE%d" % (m, int(e))
        return repr(v)
    with open(prefix + ".errorprofile", "w") as f:
        for row in P:
            f.write("".join(jd(v) + "\t" for v in row) + "\n")
    with open(prefix + ".indelprofile", "w") as f:
        f.write(jd(ins) + "\t" + jd(dele))
    return prefix + ".errorprofile", prefix + ".indelprofile"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads of the job (--scaling strong: shared by the ranks; weak: per GPU)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong", help="N>1: strong = --reads in all, contiguous range per rank (BASELINE configs[3]); weak = --reads per GPU")
    ap.add_argument("--drain", type=int, default=1, help="N=1: also time single launches of 1/8, 1/4, 1/2 of the batch (what a rank of a 8/4/2-GPU strong-scaling job would run)")
    ap.add_argument("--read-len", type=int, default=50)
    ap.add_argument("--genome-mbp", type=int, default=3100)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--genome-profile", choices=["default", "repeats"], default="default",
                    help="repeats: ~45 %% of the genome in interspersed repeat families (300-bp and 6-kb units, 2-20 %% diverged), microsatellites and "
                         "satellite arrays (add_repeats) instead of an i.i.d. text with 4 %% two-copy segments")
    ap.add_argument("--workload", choices=["full", "exact"], default="full")
    ap.add_argument("--cpu-sample", type=int, default=250000, help="reads of the same workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--penalty", choices=["profile", "stock"], default="profile", help="full workload: PAR-CLIP error-profile costs (bwa parasuite) or stock costs (bwa aln -n 0.04)")
    ap.add_argument("--sub-batches", type=int, default=1, help="2: the batch of a step is mapped as two halves on two lanes (streams) driven by two threads; measured slower than 1 (DESIGN.md section 5)")
    ap.add_argument("--pipeline", type=int, default=0, help="batches in flight per GPU (1 to 3; more than 1 needs --sub-batches 1): consecutive steps alternate between "
                    "that many batches, each on its own stream with its own 69-GB workspace, and the search launch of step k+1 is submitted while step k's still runs, "
                    "so that it takes the CUs step k's retiring workgroups leave (tools/backfill_probe.py, tools/pipeline_probe.py).  0 (default) = 2: three in "
                    "flight were measured slower than two at every batch size (1.25 M reads: 280 against 232 ms per step; 10 M: 1230 against 1228)")
    ap.add_argument("--stagger-ms", type=float, default=-1.0, help="pipelined steps: how long a search launch has the GPU to itself before the next step is started (default: 50 ms per 10 M reads)")
    ap.add_argument("--e2e", type=int, default=1, help="N=1: also time one ps_map call, FASTQ file -> closed SAM file (the reference's own timer scope)")
    ap.add_argument("--dump-hits", default="", help="directory: every rank saves the per-read hit records of its last step (tests)")
    ap.add_argument("--keep", default="")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use GPU 0")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the N>1 path (index staged through the host)")
    args = ap.parse_args()

    import threading
    import torch
    import capi
    import sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
    threads = args.threads or min(16, max(1, (os.cpu_count() or 8) // max(1, world)))
    log = (lambda *a: print("[bench r%d]" % rank, *a, file=sys.stderr, flush=True))
    S = 2 if args.sub_batches >= 2 else 1
    PIPE = 0                                                   # set below, once the rank's share of the reads is known
    do_e2e = bool(args.e2e) and world == 1

    # ---------------- data: genome on every rank (same seed), index built on rank 0 ----------------
    t0 = time.time()
    contigs = gen_genome(torch, dev, args.genome_mbp * 1_000_000, args.contigs, 0x5EED0002)
    if args.genome_profile == "repeats":
        contigs = add_repeats(torch, dev, contigs, 0x5EED0009)
    tmpdir = args.keep or tempfile.mkdtemp(prefix="psbench_")
    os.makedirs(tmpdir, exist_ok=True)
    fa = os.path.join(tmpdir, "genome.fa")
    ctx = None
    if rank == 0:
        write_fasta(fa, contigs)
        log("genome %.1f Mbp written in %.1fs" % (args.genome_mbp, time.time() - t0))
        t1 = time.time()
        torch.cuda.empty_cache()                    # the suffix sorter wants most of the HBM for a genome of this size
        ctx = capi.Ctx.build(fa, device=local, save_files=do_e2e)      # the index files are what the e2e leg's ps_map loads
        info = ctx.info()
        log("index built in %.1fs (library %.0f ms, %d doubling rounds, %.2f GB in HBM)%s" %
            (time.time() - t1, info.build_ms, info.sa_rounds, info.device_bytes / 1e9, "; files saved" if do_e2e else ""))
    if world > 1:
        # one-off broadcast of the index blobs over xGMI (RCCL); rank 0 copies device-to-device into the send buffers
        t1 = time.time()
        obj = [None]
        if rank == 0:
            obj = [(ctx.meta(), [ctx.blob(i)[1] for i in range(3)])]
        dist.broadcast_object_list(obj, src=0)
        meta, sizes = obj[0]
        blobs = [torch.empty(s, dtype=torch.uint8, device=dev) for s in sizes]
        if rank == 0:
            for i in range(3):
                ctx.export_blob(i, blobs[i].data_ptr(), sizes[i])
        torch.cuda.synchronize()
        for i, b in enumerate(blobs):
            if args.backend == "nccl":
                dist.broadcast(b, src=0)
            else:                                   # gloo rehearsal: stage through host memory
                hb = b.cpu()
                dist.broadcast(hb, src=0)
                blobs[i].copy_(hb)
        torch.cuda.synchronize()
        if rank != 0:
            ctx = capi.Ctx.from_blobs(meta, local, [b.data_ptr() for b in blobs], keep=blobs)
        log("index broadcast %.2f GB in %.2fs" % (sum(sizes) / 1e9, time.time() - t1))
    P = np.array(PROFILE)
    P[3, 1], P[3, 3] = 0.12, 0.87           # the T->C rate a first mapping pass of PAR-CLIP data yields
    if args.workload == "exact":
        ctx.set_stock("0")
    elif args.penalty == "stock":
        ctx.set_stock("0.04")
    else:
        ctx.set_profile(P, INS_RATE, DEL_RATE, -1)

    t1 = time.time()
    strong = args.scaling == "strong"
    if strong:          # one job: every rank generates the same reads (same seed) and keeps its contiguous range
        lo, hi = sharding.shard_range(args.reads, rank, world)
        codes = gen_reads(torch, dev, contigs, args.reads, args.read_len, 0x5EED0003, indels=(args.workload == "full"))[lo:hi]
    else:
        codes = gen_reads(torch, dev, contigs, args.reads, args.read_len, 0x5EED0003 + rank, indels=(args.workload == "full"))
    n_mine = codes.shape[0]
    PIPE = 1 if S > 1 else max(1, min(3, args.pipeline if args.pipeline > 0 else 2))     # lanes of work: stream + workspace each
    del contigs
    torch.cuda.empty_cache()
    # the batch of a step as S sub-batches (contiguous halves, input order) on S lanes: stream + workspace each
    ctx.set_lanes(S * PIPE)
    cut = [n_mine * j // S for j in range(S + 1)]
    sets = [[ctx.batch_from_codes(codes[cut[j]:cut[j + 1]]) for j in range(S)] for _ in range(PIPE)]    # PIPE copies of the step's batch
    batches = sets[0]
    log("%d reads (%s scaling: %d of the job's) generated, packed and uploaded as %d sub-batch(es), %d batch(es) in flight, in %.1fs" % (n_mine, args.scaling, args.reads if strong else world * args.reads, S, PIPE, time.time() - t1))

    chain = torch.zeros(1, dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
    wall = {"search": 0.0, "select_hard": 0.0, "select_easy": 0.0, "locate": 0.0, "bt_union": 0.0}

    class Step:
        """one pass of the hot path over the whole batch, in three parts.  start(): every sub-batch gets a thread (ctypes calls
        release the GIL) that runs search -> [waits for its turn in the tie-break stream] -> bulk selection -> SA walk, MAPQ, DP.
        chain(): on the CALLING thread (always the main one: every torch.distributed call of this program is made there, in step
        order) the tie-break stream goes through this rank's sub-batches, then on to the next rank.  finish(): joins, returns the
        sub-batches' timings."""

        def __init__(self, batches):
            self.batches = batches
            self.searched = [threading.Event() for _ in range(S)]
            self.chosen = [threading.Event() for _ in range(S)]
            self.err = []
            self.tw = [dict() for _ in range(S)]
            self.th = []
            self.submitted = None

        def _lane(self, j):
            b, tw = self.batches[j], self.tw[j]
            try:
                t0 = time.perf_counter()
                b.search()
                tw["search"] = time.perf_counter() - t0
                self.searched[j].set()
                self.chosen[j].wait()
                if self.err:
                    return
                t1 = time.perf_counter()
                b.select_easy(threads)
                t2 = time.perf_counter()
                b.locate()
                tw["select_easy"] = t2 - t1; tw["locate"] = time.perf_counter() - t2
            except Exception as e:           # noqa: BLE001 -- reported by finish(), the other lane is released
                self.err.append(e)
                self.searched[j].set()

        def start(self):
            self.th = [threading.Thread(target=self._lane, args=(j,)) for j in range(S)]
            for t in self.th:
                t.start()
            self.submitted = time.perf_counter()

        def chain(self):
            def advance(before):             # the tie-break stream through this rank's sub-batches, in input order
                for j in range(S):
                    self.searched[j].wait()
                    if self.err:
                        break
                    t0 = time.perf_counter()
                    before = self.batches[j].select_hard(before)
                    self.tw[j]["select_hard"] = time.perf_counter() - t0
                    self.chosen[j].set()
                return before
            try:
                sharding.chain_stream_position(dist, rank, world, chain, advance)
            finally:
                for e in self.chosen:
                    e.set()

        def finish(self):
            for e in self.chosen:
                e.set()
            for t in self.th:
                t.join()
            if self.err:
                raise self.err[0]
            tms = [b.timing() for b in self.batches]
            for k in ("search", "select_hard", "select_easy", "locate"):
                wall[k] += max(self.tw[j].get(k, 0.0) for j in range(S))
            wall["bt_union"] += (max(t["bt_end_ms"] for t in tms) - min(t["bt_begin_ms"] for t in tms)) * 1e-3
            return tms

    def run_steps(n):
        """n steps, results in step order.  With PIPE > 1 consecutive steps alternate between PIPE batches: step k+1 (.. k+PIPE-1) is
        started -- its search launch submitted -- as soon as its batch is free (step k+1-PIPE finished) and the launch before it has
        had the GPU to itself for a moment (two launches submitted together share the CUs from the start), i.e. while step k's kernel
        still runs: its workgroups start where step k's retire, and step k's short later stages run inside that hand-over."""
        out = []
        if PIPE == 1:
            for _ in range(n):
                st = Step(sets[0]); st.start()
                try:
                    st.chain()
                finally:
                    out.append(st.finish())
            return out
        steps = [None] * n
        done = [False] * n
        stagger = max(0.002, 0.05 * n_mine / 10e6)       # 50 ms behind a 10 M-read launch, in proportion for smaller batches
        if args.stagger_ms >= 0:
            stagger = args.stagger_ms * 1e-3

        def begin(k):
            if k >= PIPE and not done[k - PIPE]:
                out_k = steps[k - PIPE].finish(); done[k - PIPE] = True      # the batch is free when its previous step is through
                results[k - PIPE] = out_k
            if k > 0:
                time.sleep(max(0.0, steps[k - 1].submitted + stagger - time.perf_counter()))
            steps[k] = Step(sets[k % PIPE]); steps[k].start()
        results = [None] * n
        try:
            started = 0
            for k in range(n):
                while started < min(n, k + PIPE):      # PIPE - 1 further steps are under way while step k takes its turn in the tie-break chain
                    begin(started); started += 1
                steps[k].chain()
            for k in range(n):
                if not done[k]:
                    results[k] = steps[k].finish(); done[k] = True
        except BaseException:
            for k in range(n):               # release whatever still waits
                if steps[k] is not None and not done[k]:
                    try:
                        steps[k].finish()
                    except Exception:        # noqa: BLE001
                        pass
            raise
        return results

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if PIPE > 1:
        # set-up, not a step: the second batch in flight gets its search workspace (69 GB, allocated at a batch's first search) --
        # with --warmup 1 only the first batch would have one and the allocation would land in the timed region
        for bs in sets[1:]:
            for b in bs:
                b.search()
    run_steps(args.warmup)
    sync()
    for k in wall:
        wall[k] = 0.0
    acc = {}
    t_start = time.perf_counter()
    for tms in run_steps(args.steps):
        for tm in tms:
            for k, v in tm.items():
                acc[k] = acc.get(k, 0) + v
    sync()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.dump_hits:
        os.makedirs(args.dump_hits, exist_ok=True)
        last = sets[(max(1, args.steps) - 1) % PIPE]          # the batch the last timed step ran on
        np.save(os.path.join(args.dump_hits, "hits_rank%d.npy" % rank), np.concatenate([b.hits() for b in last]))
    if rank == 0:
        K = max(1, args.steps)
        # ---- counters: the timed kernel carries none.  One more, untimed pass of the search stage with the counting kernel,
        # one sub-batch at a time (nothing overlaps: these launch durations are the kernel alone); the counts of a batch are the
        # same in every pass (the search is deterministic)
        ctx.set_stats(True)
        cap_was = os.environ.get("PS_CAP")
        os.environ["PS_CAP"] = "0"                  # the counting pass takes the reference's steps: every child upstream stores is stored (the timed kernel leaves
        ks_bt, ks_w, solo_ms = {}, {}, 0.0         # some out on the strength of the estimated best score and searches ~0.002 % of the reads twice: ps_narrow.h, nt_tail)
        for b in batches:
            b.search()
            solo_ms += b.timing()["ms_backtrack"]
            for dst, which in ((ks_bt, 1), (ks_w, 0)):
                for k, v in b.kstats(which).items():
                    dst[k] = dst.get(k, 0) + v
        ctx.set_stats(False)
        if cap_was is None:
            del os.environ["PS_CAP"]
        else:
            os.environ["PS_CAP"] = cap_was
        solo_timed_ms = 0.0
        ks_timed = {}
        for b in batches:                   # leave the batches searched with the timed kernel, selected and located (hits below);
            b.run(threads)                  # this pass is also the timed kernel ALONE on the GPU: nothing else is in flight
            solo_timed_ms += b.timing()["ms_backtrack"]
            for k, v in b.kstats(1).items():    # the timed kernel counts the steps it takes through the Occ array (the others: jump table)
                ks_timed[k] = ks_timed.get(k, 0) + v
        ks_sa = {}
        for b in batches:
            for k, v in b.kstats(2).items():
                ks_sa[k] = ks_sa.get(k, 0) + v
        # ---- the same steps one after the other (what --pipeline 1 times): nothing overlaps, so the HIP events around each
        # launch are the kernel's own duration INSIDE a timed region; rocprofv3 --kernel-trace of `bench.py --pipeline 1` agrees
        pipe1 = None
        if PIPE > 1 and world == 1:
            walls, kms = [], []
            for _ in range(3):
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for b in batches:
                    b.run(threads)
                torch.cuda.synchronize()
                walls.append(1e3 * (time.perf_counter() - t2)); kms.append(sum(b.timing()["ms_backtrack"] for b in batches))
            pipe1 = {"ms_per_step": sum(walls) / len(walls), "kernel_ms_per_step": sum(kms) / len(kms), "steps": len(walls),
                     "reads_per_s": n_mine / (sum(walls) / len(walls) * 1e-3)}
        # ---- launch drain (N=1): what a rank of a 2/4/8-GPU strong-scaling job would run -- the first 1/2, 1/4, 1/8 of the batch
        # as ONE step each (search + samse stages), nothing else on the GPU.  Every search launch ends with its longest reads;
        # T(n) is not proportional to n, and that -- not the index broadcast, not the chain -- is what bounds strong scaling.
        drain = None
        if world == 1 and args.drain and S == 1:
            rows = []
            for div in (64, 8, 4, 2, 1):
                nb = max(1, n_mine // div)
                sb = batches[0] if div == 1 else ctx.batch_from_codes(codes[:nb])
                w_, k_ = [], []
                for _ in range(2):
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    sb.run(threads)
                    torch.cuda.synchronize()
                    w_.append(1e3 * (time.perf_counter() - t2)); k_.append(sb.timing()["ms_backtrack"])
                rows.append({"reads": nb, "ms_step": min(w_), "ms_backtrack": min(k_)})
                if div != 1:
                    sb.free()
            full = rows[-1]["ms_step"]
            drain = {"single_launch_steps": rows,
                     "implied_speedup": {str(g): full / r["ms_step"] for g, r in zip((8, 4, 2), rows[1:4])},
                     "floor": "the first row holds fewer reads than the launch has lanes (262,144): every read has a lane to itself from the start, and the launch "
                              "still takes ms_backtrack -- the time of its LONGEST search (up to 126,000 dependent iterations of ~2.5 us for a wave that has "
                              "the SIMD to itself; the best-first order makes a read's iterations sequential).  No hand-out order or batch size gets a "
                              "launch below that; what hides it is the next launch (--pipeline 2), which is how the timed steps run",
                     "note": "one step (search + samse stages) over the first 1/64, 1/8, 1/4, 1/2 and all of the batch, each alone on the GPU, best of two; "
                             "implied_speedup[G] = T(all) / T(1/G): what G GPUs can reach on this job (--scaling strong) before any multi-GPU cost"}
        n_bt = max(1, acc["n_backtrack_launches"])
        ms_bt_sum_step = acc["ms_backtrack"] / K          # summed over the launches of a step (two lanes overlap in time)
        ms_bt_union_step = 1e3 * wall["bt_union"] / K       # first launch's start to last launch's end: the time the kernel had the GPU
        if PIPE > 1:
            # launches of consecutive steps overlap (a launch is submitted while its predecessor runs and its events bracket the
            # wait for free CUs too): the kernel's own duration is taken from the solo pass above, same kernel, same batch
            ms_bt_union_step = solo_timed_ms
            ms_bt_sum_step = solo_timed_ms
        ms_w_step = acc["ms_width"] / K
        # algorithmic bytes: 64 B x distinct Occ blocks touched by the search steps
        alg_bt = 64.0 * (2 * ks_bt["occ_pairs"] - ks_bt["occ_same_blk"])
        alg_w = 64.0 * (2 * ks_w["occ_pairs"] - ks_w["occ_same_blk"])
        dominant_bt = ms_bt_union_step >= ms_w_step
        launches_per_step = (n_bt if dominant_bt else max(1, acc["n_width_launches"])) / K
        ach = (alg_bt / (ms_bt_union_step * 1e-3) if dominant_bt else alg_w / (ms_w_step * 1e-3)) / 1e9
        hits = np.concatenate([b.hits() for b in batches])
        res = {
            "metric": "aligned reads/sec (10Mx50bp PAR-CLIP vs hg19-size genome) on MI355X; SAM bit-exact vs own CPU restatement (parity unpinned)",
            "value": (args.reads if strong else world * args.reads) * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: %dx%dbp simulated PAR-CLIP reads %s, %s, vs %d Mbp synthetic genome "
                                   "(hg19-size synthetic genome; 33-bit BWT rows)" % (
                                   "configs[2]" if world == 1 else ("configs[3]" if strong else "configs[2] per GPU"), args.reads, args.read_len,
                                   "in all, contiguous ranges of ceil(reads/%d) per GPU" % world if strong else "per GPU",
                                   "error-profile seed + banded extension" if args.workload == "full" else "exact-match seed only",
                                   args.genome_mbp),
                       "reads_total": args.reads if strong else world * args.reads,
                       "reads_per_gpu": -(-args.reads // world) if strong else args.reads, "read_len": args.read_len, "genome_mbp": args.genome_mbp,
                       "genome_profile": args.genome_profile,
                       "mode": args.workload, "penalty": args.penalty, "sub_batches": S, "pipeline": PIPE,
                       "parallelism": "reads sharded x%d, index replicated" % world},
            "value_scope": "search + samse stages on reads already packed in HBM -> per-read alignment records in pinned host memory "
                           "(FASTQ parsing, upload and SAM text are outside; see value_e2e)",
            "roofline": {"bound": "hbm", "kernel": "k_backtrack_n" if dominant_bt else "k_width",
                         "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None, "traffic_source": None,
                         "achieved_definition": "ALGORITHMIC bytes of the reference's algorithm (SURVEY.md 8d: 64 B per distinct Occ block of every search step, counted by "
                                                "an untimed pass of the counting kernel WITHOUT the jump table) / the timed kernel's duration.  The timed kernel asks the "
                                                "memory for less (requested_GBps: its own counters, 32-byte table slots for the shallow steps); measured HBM traffic is `traffic`",
                         "algorithmic_GBps": ach,
                         "algorithmic_bytes_per_step": (alg_bt if dominant_bt else alg_w),
                         "algorithmic_bytes_per_launch": (alg_bt if dominant_bt else alg_w) / launches_per_step,
                         "launches_per_step": launches_per_step,
                         "kernel_ms_per_step_union": ms_bt_union_step if dominant_bt else ms_w_step,
                         "kernel_ms_per_step_sum_of_launches": ms_bt_sum_step if dominant_bt else ms_w_step,
                         "kernel_ms_per_step_launches_alone": solo_timed_ms if dominant_bt else ms_w_step,
                         "counting_kernel_ms_per_step": solo_ms,
                         "avg_launch_ms": (ms_bt_sum_step if dominant_bt else ms_w_step) / launches_per_step,
                         "timing": ("HIP events on the launch's own stream.  Steps are pipelined (config.pipeline = 2): the launch of step k+1 is in the queue "
                                    "while step k's kernel runs, so events around the timed launches would include that wait; avg_launch_ms is the same "
                                    "kernel over the same batch run ALONE right after the timed region (it agrees with rocprofv3 --stats of "
                                    "`bench.py --pipeline 1`), achieved = bytes of a launch / that duration.  Because a launch fills the CUs its "
                                    "predecessor's retiring workgroups leave, ms_per_step can be shorter than one launch alone"
                                    if PIPE > 1 else
                                    "HIP events on each lane's own stream around every launch of the timed steps; with two sub-batches the lanes' launches "
                                    "overlap and achieved = bytes of a step / (first launch's start to last launch's end)") +
                                   "; counters from one extra untimed pass of the counting kernel over the same batch",
                         "requested_bytes_per_step": (64.0 * (2 * ks_timed.get("occ_pairs", 0) - ks_timed.get("occ_same_blk", 0)) +
                                                      32.0 * max(0, ks_bt["occ_pairs"] - ks_timed.get("occ_pairs", 0))) if dominant_bt else None,
                         "requested_GBps": ((64.0 * (2 * ks_timed.get("occ_pairs", 0) - ks_timed.get("occ_same_blk", 0)) +
                                             32.0 * max(0, ks_bt["occ_pairs"] - ks_timed.get("occ_pairs", 0))) / (ms_bt_union_step * 1e-3) / 1e9) if dominant_bt else None,
                         "requested_bytes_note": "what the timed kernel asks the memory for in search steps: 64 B per distinct Occ block of the steps it takes through "
                                                 "the Occ array (its own two counters) + 32 B per step answered by the jump table (DESIGN.md section 2); the algorithmic "
                                                 "bytes above are those of the reference's algorithm, counted without the table",
                         "width_kernel": {"achieved": alg_w / (ms_w_step * 1e-3) / 1e9, "frac": alg_w / (ms_w_step * 1e-3) / 1e9 / 8000.0},
                         "pipeline1": (dict(pipe1, achieved=alg_bt / (pipe1["kernel_ms_per_step"] * 1e-3) / 1e9,
                                            frac=alg_bt / (pipe1["kernel_ms_per_step"] * 1e-3) / 1e9 / 8000.0,
                                            note="the same batch, steps one after the other right after the timed region (what `bench.py --pipeline 1` times): "
                                                 "HIP events around every launch, inside the region, nothing overlapping") if pipe1 and dominant_bt else None)},
            "kernels_ms_per_step": {k: acc[k] / K for k in ("ms_width", "ms_backtrack", "ms_select", "ms_sa2pos",
                                                            "ms_refine", "ms_host_post", "ms_classify", "ms_sel_hard", "ms_sel_easy")},
            "stage_wall_ms_per_step": {k: 1e3 * v / K for k, v in wall.items()},
            "note_pipelined_timings": ("config.pipeline = 2: the per-stage and per-kernel times above are taken on steps that overlap in time -- ms_backtrack "
                                       "there brackets the launch's wait for free CUs as well, stage walls add up to more than ms_per_step; the kernel alone is "
                                       "roofline.avg_launch_ms") if PIPE > 1 else None,
            "kstats": {"backtrack": ks_bt, "width": ks_w, "sa2pos": ks_sa, "backtrack_timed_kernel": ks_timed},
            "mapped_frac": float((hits["type"] != 0).mean()),
            "x0_gt1_frac": float((hits["c1"] > 1).mean()), "x0_gt30_frac": float((hits["c1"] > 30).mean()),
            "overflow_reads": [int(acc["n_overflow_tier1"] / K), int(acc["n_overflow_tier2"] / K)],
            "drain": drain,
        }
        # HBM traffic of the dominant kernel: PMC passes cannot run inside the timed job, so the committed passes of
        # the same configuration are quoted (profiles/pmc_traffic.json), null when none matches
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            for e in pt["entries"]:
                if all(res["config"].get(k) == v for k, v in e["match"].items()) and e["kernel"] == res["roofline"]["kernel"]:
                    res["roofline"]["traffic"] = (e["fetch_kb"] + e["write_kb"]) * 1024.0 / e["launches"] * launches_per_step
                    res["roofline"]["traffic_source"] = "from committed profile, not this run: " + e["source"]
        except Exception:
            pass
        # ---------------- CPU baseline: the oracle (a port, not the PARA-suite_aligner binary) ----------------
        if world == 1 and args.cpu_sample > 0:
            try:
                import orc
                import simulate as S_
                t2 = time.time()
                ns = min(args.cpu_sample, args.reads)
                sim = dict(codes=codes[:ns], lens=np.full(ns, args.read_len, dtype=np.int32),
                           quals=np.full((ns, args.read_len), 73, dtype=np.uint8))
                fq = os.path.join(tmpdir, "sample.fq")
                S_.write_fastq(fq, sim, names=["r%d" % i for i in range(ns)])
                info = ctx.info()
                oix = orc.Index.from_parts(fa, ctx.bwt_syms_chunked(), info.primary, ctx.sa_samples())
                if args.workload == "exact":
                    oopt = orc.stock_opt("0")
                elif args.penalty == "stock":
                    oopt = orc.stock_opt("0.04")
                else:
                    oopt = orc.profile_opt(P, INS_RATE, DEL_RATE, -1)
                cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
                cores = min(cores, int(os.environ.get("PS_CPU_THREADS", "16")))   # the GPU box grants 16 host cores per GPU
                log("oracle index adopted in %.1fs; timing %d reads on %d threads" % (time.time() - t2, ns, cores))
                osam = os.path.join(tmpdir, "sample.orc.sam")
                r = oix.map_fastq(oopt, fq, osam, n_threads=cores)
                # two scopes, like for like with the GPU figures: up to the per-read records (value), and file to file (value_e2e)
                cpu_rate = ns / (r["t_aln"] + r["t_samse_records"])
                cpu_rate_e2e = ns / (r["t_parse"] + r["t_aln"] + r["t_samse"])
                res["cpu_baseline"] = {"value": cpu_rate, "unit": "reads/s", "cores": cores, "kind": "port",
                                       "scope": "aln + samse stages up to the per-read alignment records (no FASTQ parse, no SAM text): the scope of `value`",
                                       "value_e2e": cpu_rate_e2e,
                                       "scope_e2e": "FASTQ parse + aln + samse + SAM text written (index already in memory): the scope of `value_e2e` minus the index load",
                                       "seconds": {k: r[k] for k in ("t_parse", "t_aln", "t_samse_records", "t_sam_text")},
                                       "sample": "first %d reads of the same batch; CPU restatement (oracle/ps_oracle.c, OpenMP), not the "
                                                 "PARA-suite_aligner binary; the index it searches is adopted from the product (see tests for the "
                                                 "independent index checks)" % ns}
                # parity of the same sample through the product
                sb = ctx.batch_from_codes(codes[:ns])
                sb.run(threads)
                gsam = os.path.join(tmpdir, "sample.gpu.sam")
                sb.write_sam(gsam, header=False, threads=threads)
                def strip(path):          # drop QNAME and QUAL (the device batch was built from codes, without qualities)
                    out = []
                    for l in open(path):
                        if not l.startswith("@"):
                            f = l.rstrip("\n").split("\t")
                            out.append("\t".join(f[1:10] + f[11:]))
                    return out
                g_l, o_l = strip(gsam), strip(osam)
                # the RNG stream position differs between "first ns reads alone" and the oracle's identical run: both start at 0
                res["parity_sample"] = {"reads": ns, "identical_sam_lines": int(sum(a == b for a, b in zip(g_l, o_l))),
                                        "all_identical": g_l == o_l, "against": "own CPU restatement (parity unpinned)"}
                sb.free()
            except Exception as e:  # the baseline is a reported extra; never hide the GPU result
                res["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        # ---------------- T_e2e: the reference's own timer scope (PARAsuiteMapping.java:57,94-97) ----------------
        # index load from its files + FASTQ parse + search + samse + SAM text written and closed: ONE ps_map call, the entry point
        # a JNI / argv binding calls.  The staged objects above are released first (ps_map brings its own context).
        if do_e2e:
            try:
                for bs in sets:
                    for b in bs:
                        b.free()
                ctx.close()
                torch.cuda.empty_cache()
                t2 = time.time()
                fq_all = os.path.join(tmpdir, "reads.fq")
                write_fastq_fast(fq_all, codes)
                if args.workload == "full" and args.penalty == "profile":
                    ep, ip = write_java_profile(os.path.join(tmpdir, "bench"), P, INS_RATE, DEL_RATE)
                    mm = "-1"
                else:
                    ep, ip, mm = None, None, ("0" if args.workload == "exact" else "0.04")
                log("FASTQ (%.2f GB) written in %.1fs; ps_map ..." % (os.path.getsize(fq_all) / 1e9, time.time() - t2))
                out_sam = os.path.join(tmpdir, "reads.sam")
                # This process has just given ~150 GB of HBM back (two staged batches with their stack workspaces, the context).  The
                # driver clears freed memory in the background and an allocation that is handed such memory waits for it
                # (tools/microbench_malloc, profiles/r03_malloc_microbench.txt: 69 GB in 0.03 s on a quiet card, 2-4 s right after a
                # free of that size) -- an artefact of what THIS program did a moment ago, not of ps_map: let it finish first.
                time.sleep(6.0)
                times = []
                for rep in range(3):                     # the first call also pays one-off costs of a new process context
                    t3 = time.perf_counter()
                    capi.ps_map(threads, mm, ep, ip, fa, fq_all, out_sam)
                    times.append(time.perf_counter() - t3)
                res["t_e2e_s"] = min(times)               # best of the three calls of this process (all of them in e2e.seconds_per_call; a call that follows a large hipFree can wait seconds for the driver to clear memory)
                res["t_e2e_first_call_s"] = times[0]      # the first call of this process
                res["value_e2e"] = args.reads / res["t_e2e_s"]
                res["e2e"] = {"scope": "one ps_map call: index files -> HBM, FASTQ file parsed, search + samse, SAM text written and closed "
                                       "(the scope of the reference's own timer, PARAsuiteMapping.java:57,94-97)",
                              "seconds_per_call": times, "fastq_bytes": os.path.getsize(fq_all), "sam_bytes": os.path.getsize(out_sam),
                              "host_threads": threads}
                # ---- cold: what the UNMODIFIED jar's timer would show -- two fresh `bwa` processes per pass, exactly the argv of
                # PARAsuiteMapping.java:63-77 / 85-92 (BWAMapping.java:51-75 for stock costs), through the argv shim.  New child
                # processes (never a re-exec of this one); this process holds nothing on the GPU any more.
                try:
                    import hashlib
                    import subprocess
                    bwa = os.path.join(ROOT, "para-suite_amd", "bin", "bwa")
                    sai, cold_sam = os.path.join(tmpdir, "reads.sai"), os.path.join(tmpdir, "reads.cold.sam")
                    if ep:
                        argv1 = [bwa, "parasuite", "-t", str(threads), "-X", mm, "-p", ep, "-g", ip, fa, fq_all, "-f", sai]
                    else:
                        argv1 = [bwa, "aln", "-t", str(threads), "-n", mm, fa, fq_all, "-f", sai]
                    argv2 = [bwa, "samse", fa, sai, fq_all, "-f", cold_sam]
                    # the last warm call has just handed its 74 GB back: the same pause as in front of the warm calls (in the Java's flow the
                    # SAM -> BAM conversions of the pass before, many seconds, lie between two mapping processes)
                    time.sleep(6.0)
                    t3 = time.perf_counter()
                    subprocess.run(argv1, check=True, timeout=600)
                    t4 = time.perf_counter()
                    subprocess.run(argv2, check=True, timeout=600)
                    t5 = time.perf_counter()

                    def md5(path):
                        h = hashlib.md5()
                        with open(path, "rb") as f:
                            for blk in iter(lambda: f.read(1 << 24), b""):
                                h.update(blk)
                        return h.hexdigest()
                    res["t_e2e_cold_s"] = t5 - t3
                    res["e2e"]["cold"] = {"seconds": {"bwa parasuite|aln": t4 - t3, "bwa samse": t5 - t4},
                                          "argv": [" ".join(os.path.basename(a) if a.startswith("/") else a for a in v) for v in (argv1, argv2)],
                                          "sam_md5": md5(cold_sam), "same_bytes_as_warm_call": md5(cold_sam) == md5(out_sam),
                                          "note": "two child processes started one after the other as Mapping.executeCommand does (Mapping.java:151-198); "
                                                  "each pays process start, HIP initialisation, the index load and every allocation"}
                    os.remove(cold_sam)
                except Exception as e:  # noqa: BLE001
                    res["t_e2e_cold_s"] = None
                    res["e2e"]["cold"] = {"failed": repr(e)}
                for pth in (fq_all, out_sam):
                    os.remove(pth)
            except Exception as e:  # noqa: BLE001
                res["t_e2e_s"] = None
                res["value_e2e"] = None
                res["e2e"] = {"failed": repr(e)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
