"""T_e2e of the file-level ABI (what the Java times, PARAsuiteMapping.java:57,94-97): index load + map + SAM written."""
import sys, os, time, numpy as np
sys.path.insert(0, 'para-suite_amd'); sys.path.insert(0, '.')
import capi, torch, bench, simulate as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
mbp = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device('cuda', 0)
contigs = bench.gen_genome(torch, dev, mbp * 1_000_000, 8, 0x5EED0002)
fa = '/tmp/e2e.fa'; fq = '/tmp/e2e.fq'
bench.write_fasta(fa, contigs)
rd = bench.gen_reads(torch, dev, contigs, n, 50, 0x5EED0003, indels=True)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
t = time.time()
with open(fq, 'wb') as f:                      # fixed-width records: vectorised FASTQ writer
    name = np.char.zfill(np.arange(n).astype(str), 9).astype('S9').view(np.uint8).reshape(n, 9)
    rec = np.empty((n, 1 + 9 + 1 + 50 + 3 + 50 + 1), dtype=np.uint8)
    rec[:, 0] = ord('@'); rec[:, 1:10] = name; rec[:, 10] = 10; rec[:, 11:61] = lut[rd]; rec[:, 61] = 10; rec[:, 62] = ord('+'); rec[:, 63] = 10
    rec[:, 64:114] = ord('I'); rec[:, 114] = 10
    f.write(rec.tobytes())
print('fastq written %.1fs' % (time.time() - t))
t = time.time(); capi.ps_index(fa); print('ps_index %.1fs' % (time.time() - t))
P = np.array(bench.PROFILE); P[3, 1], P[3, 3] = 0.12, 0.87
with open('/tmp/e2e.errorprofile', 'w') as f:
    for row in P: f.write(''.join(repr(float(v)) + '\t' for v in row) + '\n')
open('/tmp/e2e.indelprofile', 'w').write('2.1E-5\t5.9E-4')
workers = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1]      # PS_WORKERS_PER_GPU settings to time
# further arguments: one run pair per argument, each a comma-joined list of NAME=value settings for the library (after the plain runs)
configs = [dict(PS_WORKERS_PER_GPU=str(w)) for w in workers] + [dict(kv.split('=') for kv in a.split(',')) for a in sys.argv[5:]]
for cfg, rep in [(c, r) for c in configs for r in range(2)]:
    os.environ.update(cfg)
    print(' '.join('%s=%s' % kv for kv in cfg.items()), flush=True)
    t = time.time(); os.environ['PS_VERBOSE']=os.environ.get('E2E_VERBOSE','2'); capi.ps_map(16, '-1', '/tmp/e2e.errorprofile', '/tmp/e2e.indelprofile', fa, fq, '/tmp/e2e.sam'); dt = time.time() - t
    print('ps_map (index load + %d reads + SAM written) %.2fs = %.2f M reads/s, SAM %.0f MB' % (n, dt, n / dt / 1e6, os.path.getsize('/tmp/e2e.sam') / 1e6), flush=True)
    if rep == 1:
        for k in cfg: os.environ.pop(k, None)
for lvl in (() if len(sys.argv) > 4 and sys.argv[4] == 'nobam' else ('6', '1')):            # the fused route: FASTQ -> filtered BAM, no SAM text
    os.environ['PS_BAM_LEVEL'] = lvl
    for kw in (dict(min_mapq=10), dict(min_mapq=10, sort_by_coordinate=True, write_index=True)):
        t = time.time(); st = capi.ps_map_to_bam(16, '-1', '/tmp/e2e.errorprofile', '/tmp/e2e.indelprofile', fa, fq, '/tmp/e2e.fused.bam', **kw); dt = time.time() - t
        print('ps_map_to_bam %s zlib level %s: %.2fs, %d of %d records kept, BAM %.0f MB' % (kw, lvl, dt, st['n_out'], st['n_in'], st['bam_bytes'] / 1e6), flush=True)
os.environ.pop('PS_BAM_LEVEL', None)
for args in (() if len(sys.argv) > 4 and sys.argv[4] == 'nobam' else (dict(min_mapq=10), dict(min_mapq=10, sort_by_coordinate=True, write_index=True))):
    t = time.time(); st = capi.ps_sam_to_bam('/tmp/e2e.sam', '/tmp/e2e.bam', threads=16, **args); dt = time.time() - t
    print('ps_sam_to_bam %s: %.2fs, %d of %d records kept, BAM %.0f MB' % (args, dt, st['n_out'], st['n_in'], st['bam_bytes'] / 1e6), flush=True)
