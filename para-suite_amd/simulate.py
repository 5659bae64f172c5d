"""Synthetic genome + PAR-CLIP read generator (numpy, seeded, vectorised).

The reference's simulator (bin/createSimulatedPARCLIPDataset.pl) needs CPAN
Math::Random and cannot run here; this module reproduces its
*distribution and read-name format*, not its code:
  * read name `SEQ_ID:<gene>|<transcript>|<chr>|<start>|<end>|<bound>-<cluster>:<i>`
    (createSimulatedPARCLIPDataset.pl:611), parsed by
    ValidateBenchmarkStatisticsPARCLIP.java:113-118 (split on '|');
  * per-base substitution by a 4x4 profile, row = true base, read orientation
    (createSimulatedPARCLIPDataset.pl:505-548; values shaped like
    examples/simulation/example.errorprofile);
  * a fraction `bound` of reads carries T->C conversions at up to 4 chosen T
    sites with site frequencies 0.66/0.24/0.08/0.04 (example.sitefrequency:1-4,
    createSimulatedPARCLIPDataset.pl:61,284,357);
  * at most one indel per read from per-position rates (:552-570);
  * qualities int(N(mean,sd)) clipped to [3,64], Phred+33 (:621-633).
Positions are uniform over the N-free genome rather than clustered on
transcripts.
"""
import numpy as np

# P(read base | true base), rows/cols A,C,G,T -- same magnitudes as the
# reference's example profile (3 decimals), regenerated here, not copied.
EXAMPLE_PROFILE = np.array([[0.990, 0.004, 0.003, 0.003],
                            [0.004, 0.990, 0.003, 0.003],
                            [0.006, 0.010, 0.977, 0.007],
                            [0.005, 0.005, 0.003, 0.987]])
SITE_FREQ = np.array([0.66, 0.24, 0.08, 0.04])
QUAL_MEAN, QUAL_SD = 31.0, 4.2
INS_RATE, DEL_RATE = 2.1e-5, 5.9e-4      # mean per-position rates of example.indels
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def make_contig(n_bp, rng, n_runs=(), softmask_frac=0.39, at=0.295):
    """ASCII bases of one contig: i.i.d. with hg19-like composition, N runs, soft-masked blocks."""
    cdf = np.cumsum([at, 0.5 - at, 0.5 - at]).astype(np.float32)
    asc = np.empty(n_bp, dtype=np.uint8)
    for a in range(0, n_bp, 1 << 26):                     # chunked: 1 Gbp must not allocate 8-byte temporaries
        b = min(n_bp, a + (1 << 26))
        u = rng.random(b - a, dtype=np.float32)
        asc[a:b] = BASES[(u >= cdf[0]).astype(np.uint8) + (u >= cdf[1]) + (u >= cdf[2])]
    if softmask_frac > 0:
        # alternate upper/lower blocks with geometric lengths so that ~softmask_frac is lower case
        mean_blk = 300.0
        n_blk = int(n_bp / mean_blk) + 2
        lens = rng.geometric(1.0 / mean_blk, size=n_blk)
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
        low = rng.random(n_blk) < softmask_frac
        mask = np.zeros(n_bp + 1, dtype=np.int8)
        s = starts[low]; e = np.minimum(starts[low] + lens[low], n_bp)
        ok = s < n_bp
        np.add.at(mask, s[ok], 1); np.add.at(mask, e[ok], -1)
        asc[np.cumsum(mask[:-1]) > 0] |= 0x20
    for a, b in n_runs:
        asc[a:b] = ord("N")
    return asc


def example_genome(seed=0x5EED0001):
    """483,300-bp contig `chr1` shaped like examples/references/reference_chr1.fa
    (3 N runs, ~39% soft-masked) -- regenerated, not copied."""
    rng = np.random.default_rng(seed)
    runs = [(0, 10000), (207666, 257666), (297968, 347968)]
    return [("chr1", make_contig(483300, rng, runs))]


def big_genome(total_bp, n_contigs=8, seed=0x5EED0002, n_run_len=20000):
    """hg19-like scale model: n_contigs contigs, an N run at both ends and one inside each."""
    rng = np.random.default_rng(seed)
    out = []
    per = total_bp // n_contigs
    for c in range(n_contigs):
        n = per if c < n_contigs - 1 else total_bp - per * (n_contigs - 1)
        runs = []
        if n > 20 * n_run_len:
            mid = int(n * 0.4)
            runs = [(0, n_run_len // 2), (mid, mid + n_run_len), (n - n_run_len // 2, n)]
        out.append(("chr%d" % (c + 1), make_contig(n, rng, runs, softmask_frac=0.0)))
    return out


def write_fasta(path, contigs, width=50):
    with open(path, "wb") as f:
        for name, asc in contigs:
            f.write(b">" + name.encode() + b"\n")
            n = asc.size
            full = n // width * width
            if full:
                body = np.empty((full // width, width + 1), dtype=np.uint8)
                body[:, :width] = asc[:full].reshape(-1, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if full < n:
                f.write(asc[full:].tobytes() + b"\n")


def contig_codes(asc):
    lut = np.full(256, 4, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i; lut[ch | 0x20] = i
    return lut[asc]


def simulate_reads(contigs, n_reads, read_len=50, seed=0x5EED0002, bound=0.6, profile=EXAMPLE_PROFILE,
                   indel_scale=0.0, min_len=None, n_frac=0.0):
    """Returns dict(codes [n,Lmax] uint8 (pad 255), lens, quals [n,Lmax] (ASCII), names list-like bytes[n],
    truth=(contig idx, start0, end0, strand, bound)).  min_len: uniform lengths in [min_len, read_len]."""
    rng = np.random.default_rng(seed)
    L = read_len
    lens = np.full(n_reads, L, dtype=np.int32) if min_len is None else rng.integers(min_len, L + 1, size=n_reads).astype(np.int32)
    ccodes = [contig_codes(a) for _, a in contigs]
    sizes = np.array([c.size for c in ccodes], dtype=np.int64)
    # sample positions; reject windows touching N (resample a few rounds)
    span = L + 2                       # room for one deletion
    cidx = rng.choice(len(contigs), size=n_reads, p=sizes / sizes.sum())
    start = np.zeros(n_reads, dtype=np.int64)
    todo = np.arange(n_reads)
    nmask = []
    for c in ccodes:
        cs = np.concatenate([[0], np.cumsum(c == 4, dtype=np.int64)])
        nmask.append(cs)
    for _ in range(64):
        if todo.size == 0:
            break
        s = (rng.random(todo.size) * (sizes[cidx[todo]] - span)).astype(np.int64)
        bad = np.zeros(todo.size, dtype=bool)
        for ci in range(len(contigs)):
            m = cidx[todo] == ci
            if m.any():
                cs = nmask[ci]
                bad[m] = (cs[s[m] + span] - cs[s[m]]) > 0
        start[todo] = s
        todo = todo[bad]
    if todo.size:
        raise RuntimeError("could not place reads outside N runs")
    strand = rng.random(n_reads) < 0.5
    is_bound = rng.random(n_reads) < bound
    # indels: at most one per read
    has_ins = np.zeros(n_reads, dtype=bool); has_del = np.zeros(n_reads, dtype=bool)
    ipos = np.zeros(n_reads, dtype=np.int32)
    if indel_scale > 0:
        u = rng.random(n_reads)
        p_ins = 1 - (1 - INS_RATE * indel_scale) ** lens
        p_del = 1 - (1 - DEL_RATE * indel_scale) ** lens
        has_ins = u < p_ins
        has_del = (~has_ins) & (u < p_ins + p_del)
        ipos = (6 + rng.random(n_reads) * np.maximum(lens - 12, 1)).astype(np.int32)
    ref_len = lens + has_del.astype(np.int32) - has_ins.astype(np.int32)
    # gather the true bases (forward strand window of ref_len), then orient
    idx = np.arange(span, dtype=np.int64)[None, :]
    win = np.empty((n_reads, span), dtype=np.uint8)
    for ci in range(len(contigs)):
        m = np.nonzero(cidx == ci)[0]
        if m.size:
            win[m] = ccodes[ci][start[m, None] + idx]
    # reverse strand: take revcomp of window[0:ref_len]
    j = np.arange(span, dtype=np.int32)[None, :]
    rl = ref_len[:, None]
    src = np.where(strand[:, None], rl - 1 - j, j)
    src = np.clip(src, 0, span - 1)
    true = np.take_along_axis(win, src, axis=1)
    true = np.where(strand[:, None], COMP[true], true)
    # apply indel in read orientation: build index map read pos -> true pos
    jj = np.arange(L, dtype=np.int32)[None, :]
    tpos = jj + (has_del[:, None] & (jj >= ipos[:, None])).astype(np.int32) - (has_ins[:, None] & (jj > ipos[:, None])).astype(np.int32)
    tpos = np.clip(tpos, 0, span - 1)
    read = np.take_along_axis(true, tpos, axis=1)
    ins_here = has_ins[:, None] & (jj == ipos[:, None])
    read = np.where(ins_here, rng.integers(0, 4, size=(n_reads, L)).astype(np.uint8), read)
    # T->C conversion sites on bound reads (before sequencing errors)
    isT = (read == 3) & (jj < lens[:, None])
    prio = np.where(isT, rng.random((n_reads, L)), 2.0)
    order = np.argsort(prio, axis=1)[:, :4]                     # up to 4 random T sites
    site_ok = np.take_along_axis(prio, order, axis=1) < 1.5
    conv = site_ok & (rng.random((n_reads, 4)) < SITE_FREQ[None, :]) & is_bound[:, None]
    rows = np.repeat(np.arange(n_reads), 4).reshape(n_reads, 4)
    read[rows[conv], order[conv]] = 1
    # sequencing errors by profile row of the (possibly converted) base
    cdf = np.cumsum(np.asarray(profile, dtype=np.float64) / np.sum(profile, axis=1, keepdims=True), axis=1)
    u = rng.random((n_reads, L))
    newb = (u[..., None] > cdf[read][..., :3]).sum(axis=-1).astype(np.uint8)
    read = newb
    if n_frac > 0:
        read = np.where(rng.random((n_reads, L)) < n_frac, np.uint8(4), read)
    read = np.where(jj < lens[:, None], read, np.uint8(255))
    q = np.floor(rng.normal(QUAL_MEAN, QUAL_SD, size=(n_reads, L))).astype(np.int32)
    q = np.where(q > 64, 64, np.where(q <= 2, 3, q))
    quals = (q + 33).astype(np.uint8)
    end = start + ref_len
    return dict(codes=read, lens=lens, quals=quals, cidx=cidx, start=start, end=end, strand=strand,
                bound=is_bound, has_ins=has_ins, has_del=has_del, contig_names=[n for n, _ in contigs])


def read_names(sim):
    """Names in the simulator's format; start/end 1-based as the Perl emits genomic positions."""
    n = sim["lens"].size
    out = []
    cn = sim["contig_names"]
    for i in range(n):
        out.append("SEQ_ID:g%d|t%d|%s|%d|%d|%d-%d:%d" % (i // 16, i // 16, cn[sim["cidx"][i]], sim["start"][i] + 1,
                                                          sim["end"][i], 1 if sim["bound"][i] else 0, i // 16 + 1, i % 16))
    return out


def write_fastq(path, sim, names=None):
    """Vectorised FASTQ writer (fixed-width records when all reads share one length)."""
    codes, lens, quals = sim["codes"], sim["lens"], sim["quals"]
    n, L = codes.shape
    if names is None:
        names = read_names(sim)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seq = lut[np.minimum(codes, 4)]
    with open(path, "wb") as f:
        chunk = 100000
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            parts = []
            for i in range(a, b):
                l = lens[i]
                parts.append(b"@" + names[i].encode() + b"\n" + seq[i, :l].tobytes() + b"\n+\n" + quals[i, :l].tobytes() + b"\n")
            f.write(b"".join(parts))


def score_truth(sam_path, tol=5):
    """The reference's acceptance rule (ValidateBenchmarkStatisticsPARCLIP.java:145-159): same chr,
    start-5 <= alignmentStart and end+5 >= alignmentEnd.  Returns (mapped, correct, total)."""
    import re
    mapped = correct = total = 0
    for line in open(sam_path):
        if line.startswith("@"):
            continue
        f = line.split("\t")
        total += 1
        if int(f[1]) & 4:
            continue
        mapped += 1
        nm = f[0].split("|")
        rs, re_ = int(nm[3]), int(nm[4])
        span = sum(int(x) for x, op in re.findall(r"(\d+)([MIDNS])", f[5]) if op in "MDN")
        a0 = int(f[3]); a1 = a0 + span - 1
        if nm[2] == f[2] and rs - tol <= a0 and re_ + tol >= a1:
            correct += 1
    return mapped, correct, total
