#!/usr/bin/env python3
"""ktune.py -- kernel tuning sweeps on one index / one read batch (measurement aid, not a product path).

Builds the synthetic genome and index of bench.py once, then runs the search stage under a list of settings
and prints one JSON line per setting (kernel ms from HIP events, counters, overflow counts).

  python tools/ktune.py --genome-mbp 3100 --reads 10000000 --pool 512,1024,2048,16384 --env "PS_JUMP=0|1"
Settings: --pool = tier-1 stack entries per lane (ps_ctx_set_tiers), --blocks = grid of the search kernel (0 = default),
--env NAME=v1|v2,... = values of the library's PS_* knobs (read at every launch); PARASUITE_LIB=<path> runs another build.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "para-suite_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=int, default=1000)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=50)
    ap.add_argument("--penalty", default="profile", choices=["profile", "stock"])
    ap.add_argument("--pool", default="16384")
    ap.add_argument("--variant", default="")
    ap.add_argument("--blocks", default="0")
    ap.add_argument("--env", default="", help="extra NAME=v1|v2 sweeps, comma separated (e.g. PS_FETCH_MIN=4|8|16)")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--stats", type=int, default=0, help="1: search kernel with counters (kstats), not the timed kernel")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    import bench as B
    import capi

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    log = lambda *a: print("[ktune]", *a, file=sys.stderr, flush=True)
    t0 = time.time()
    contigs = B.gen_genome(torch, dev, args.genome_mbp * 1_000_000, args.contigs, 0x5EED0002)
    import tempfile
    tmp = tempfile.mkdtemp(prefix="pstune_")
    fa = os.path.join(tmp, "genome.fa")
    B.write_fasta(fa, contigs)
    torch.cuda.empty_cache()
    ctx = capi.Ctx.build(fa, device=0)
    log("genome + index in %.1fs" % (time.time() - t0))
    if args.penalty == "stock":
        ctx.set_stock("0.04")
    else:
        P = np.array(B.PROFILE)
        P[3, 1], P[3, 3] = 0.12, 0.87
        ctx.set_profile(P, B.INS_RATE, B.DEL_RATE, -1)
    codes = B.gen_reads(torch, dev, contigs, args.reads, args.read_len, 0x5EED0003)
    del contigs
    torch.cuda.empty_cache()
    ctx.set_stats(bool(args.stats))
    batch = ctx.batch_from_codes(codes)
    log("reads ready in %.1fs" % (time.time() - t0))

    pools = [int(x) for x in args.pool.split(",") if x]
    variants = [x for x in args.variant.split(",")] if args.variant else [None]
    blocks = [int(x) for x in args.blocks.split(",") if x]
    extra = []
    for kv in [x for x in args.env.split(",") if x]:
        k, v = kv.split("=")
        extra.append((k, v.split("|")))
    out = open(args.out, "a") if args.out else None

    def run(setting):
        for k, v in setting.get("env", {}).items():
            os.environ[k] = v
        ctx.set_tiers(pool_cap=[setting["pool"], 65535, 2000064], bt_blocks=setting["blocks"])
        ms, tot = [], []
        for _ in range(args.steps):
            t1 = time.perf_counter()
            batch.search()
            tot.append(1e3 * (time.perf_counter() - t1))
            ms.append(batch.timing()["ms_backtrack"])
        tm = batch.timing()
        ks = batch.kstats(1)
        rec = dict(setting=setting, ms_backtrack=ms, ms_search_wall=tot, n_launches=tm["n_backtrack_launches"],
                   overflow=[tm["n_overflow_tier1"], tm["n_overflow_tier2"]], kstats=ks, reads=args.reads, genome_mbp=args.genome_mbp)
        if ks["occ_pairs"]:
            alg = 64.0 * (2 * ks["occ_pairs"] - ks["occ_same_blk"])
            rec["alg_TBps"] = alg / (min(ms) * 1e-3) / 1e12
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n"); out.flush()

    for var in variants:
        for pc in pools:
            for bl in blocks:
                base_env = {} if var is None else {"PS_BT_VARIANT": var}
                if not extra:
                    run(dict(pool=pc, blocks=bl, env=base_env))
                for k, vals in extra:
                    for v in vals:
                        e = dict(base_env); e[k] = v
                        run(dict(pool=pc, blocks=bl, env=e))


if __name__ == "__main__":
    main()
