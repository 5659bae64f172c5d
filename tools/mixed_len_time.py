"""throughput on ragged input (BASELINE configs[4] shape: lengths uniform on 36..75) vs fixed 50 bp; profiling aid"""
import sys, os, time, numpy as np
sys.path.insert(0, 'para-suite_amd'); sys.path.insert(0, '.')
import capi, torch, bench
mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
dev = torch.device('cuda', 0)
contigs = bench.gen_genome(torch, dev, mbp * 1_000_000, 24, 0x5EED0002)
fa = '/tmp/g.fa'; bench.write_fasta(fa, contigs); torch.cuda.empty_cache()
ctx = capi.Ctx.build(fa)
P = np.array(bench.PROFILE); P[3,1], P[3,3] = 0.12, 0.87
ctx.set_profile(P, bench.INS_RATE, bench.DEL_RATE, -1)
rd = bench.gen_reads(torch, dev, contigs, n, 75, 0x5EED0005, indels=True)
rng = np.random.default_rng(5)
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (36, 75)
lens = rng.integers(lo, hi + 1, n)
# ragged FASTQ through the file-level reader (the only entry that takes mixed lengths)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
fq = '/tmp/mixed.fq'
with open(fq, 'wb') as f:
    for a in range(0, n, 500000):
        b = min(n, a + 500000)
        out = []
        for i in range(a, b):
            s = lut[rd[i, :lens[i]]].tobytes()
            out.append(b'@r%d\n%s\n+\n%s\n' % (i, s, b'I' * lens[i]))
        f.write(b''.join(out))
for tag, path in (('mixed %d-%d' % (lo, hi), fq),):
    b = ctx.batch_from_fastq(path)
    for rep in range(2):
        t = time.time(); b.run(16); dt = time.time() - t
        tm = b.timing()
        print("%s: %d reads, run %.2fs = %.2f M reads/s; backtrack %.0f ms in %d launches, width %.0f ms, tier re-runs %d/%d, mapped %.3f" % (tag, n, dt, n / dt / 1e6, tm["ms_backtrack"], tm["n_backtrack_launches"], tm["ms_width"], tm["n_overflow_tier1"], tm["n_overflow_tier2"], float((b.hits()["type"] != 0).mean())), flush=True)
    b.free()
b = ctx.batch_from_codes(np.ascontiguousarray(rd[:, :50]))
for rep in range(2):
    t = time.time(); b.run(16); dt = time.time() - t
    tm = b.timing()
    print('fixed 50: %d reads, run %.2fs = %.2f M reads/s; backtrack %.0f ms in %d launches' % (n, dt, n / dt / 1e6, tm['ms_backtrack'], tm['n_backtrack_launches']), flush=True)
