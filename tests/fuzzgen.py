"""Seeded synthetic alignments for differential tests (oracle vs HIP path).

Reads are sampled from small transcript models laid over a random contig so
that junctions collect many alignments, alignments span several junctions,
alternative isoforms put introns inside other junctions' anchor windows, and
CIGARs carry every operation the reference handles (M I D N S H P = X)."""
import numpy as np

from portcullis_amd.records import ReadBatch

BASES = "ACGT"


def random_genome(rng, n, lower_frac=0.1, n_frac=0.002, iupac_frac=0.001):
    g = rng.choice(list(BASES), size=n)
    if n_frac:
        g[rng.random(n) < n_frac] = "N"
    if iupac_frac:
        m = rng.random(n) < iupac_frac
        g[m] = rng.choice(list("RYSWKMBDHV"), size=int(m.sum()))
    s = "".join(g)
    if lower_frac:
        # lower-case (soft-masked) stretches
        arr = list(s)
        for _ in range(max(1, int(n * lower_frac / 200))):
            a = int(rng.integers(0, n))
            b = min(n, a + int(rng.integers(20, 400)))
            arr[a:b] = [c.lower() for c in arr[a:b]]
        s = "".join(arr)
    return s


def make_transcripts(rng, glen, n_tx=12, canonical_frac=0.7, genome=None):
    """Each transcript: list of exons [(start, end_exclusive)].  Some share introns, some
    retain introns (so windows of other junctions include an intron)."""
    txs = []
    for _ in range(n_tx):
        pos = int(rng.integers(50, glen // 3))
        exons = []
        for _e in range(int(rng.integers(2, 7))):
            elen = int(rng.choice([3, 8, 20, 40, 60, 120, 300]))
            if pos + elen >= glen - 50:
                break
            exons.append((pos, pos + elen))
            ilen = int(rng.choice([1, 5, 9, 12, 40, 70, 150, 400, 1500]))
            pos = pos + elen + ilen
            if pos >= glen - 400:
                break
        if len(exons) >= 2:
            txs.append(exons)
    # isoforms: copy a transcript and merge two neighbouring exons (intron retention) or shift an acceptor
    extra = []
    for ex in txs:
        if len(ex) >= 3 and rng.random() < 0.6:
            k = int(rng.integers(0, len(ex) - 1))
            merged = ex[:k] + [(ex[k][0], ex[k + 1][1])] + ex[k + 2:]
            if len(merged) >= 2:
                extra.append(merged)
        if rng.random() < 0.5:
            k = int(rng.integers(1, len(ex)))
            s, e = ex[k]
            if e - s > 6:
                alt = ex[:k] + [(s + int(rng.integers(1, 4)), e)] + ex[k + 1:]
                extra.append(alt)
    txs += extra
    if genome is not None:
        # plant canonical / semi-canonical motifs on a share of introns
        g = list(genome)
        for ex in txs:
            for (a0, a1), (b0, b1) in zip(ex[:-1], ex[1:]):
                if b0 - a1 < 4:
                    continue
                r = rng.random()
                if r < canonical_frac * 0.5:
                    d, a = "GT", "AG"
                elif r < canonical_frac:
                    d, a = "CT", "AC"
                elif r < canonical_frac + 0.1:
                    d, a = ("GC", "AG") if rng.random() < 0.5 else ("AT", "AC")
                else:
                    continue
                g[a1:a1 + 2] = list(d)
                g[b0 - 2:b0] = list(a)
        genome = "".join(g)
    return txs, genome


def _tx_read(rng, genome, exons, L, opts):
    """Sample one read of (at most) L aligned bases from a transcript; returns dict or None."""
    tlen = sum(e - s for s, e in exons)
    if tlen < 4:
        return None
    L = min(L, tlen)
    off = int(rng.integers(0, tlen - L + 1))
    # walk exons
    segs = []  # (genomic start, len)
    rem, o = L, off
    for s, e in exons:
        n = e - s
        if o >= n:
            o -= n
            continue
        take = min(n - o, rem)
        segs.append((s + o, take))
        rem -= take
        o = 0
        if rem == 0:
            break
    pos = segs[0][0]
    ops = []  # (op, len)
    seq = []
    G = genome.upper()
    for k, (gs, n) in enumerate(segs):
        if k > 0:
            prev_end = segs[k - 1][0] + segs[k - 1][1]
            ops.append(("N", gs - prev_end))
        # split the segment with indels / = / X
        r = gs
        left = n
        while left > 0:
            roll = rng.random()
            if roll < opts["indel"] and left > 4:
                a = int(rng.integers(1, left - 2))
                ops.append(("M", a)); seq.append(G[r:r + a]); r += a; left -= a
                if rng.random() < 0.5:
                    il = int(rng.integers(1, 4))
                    ops.append(("I", il)); seq.append("".join(rng.choice(list(BASES), size=il)))
                else:
                    dl = int(min(rng.integers(1, 4), left - 1))
                    ops.append(("D", dl)); r += dl; left -= dl
            elif roll < opts["indel"] + opts["eqx"]:
                a = int(rng.integers(1, left + 1))
                ops.append((str(rng.choice(["=", "X"])), a)); seq.append(G[r:r + a]); r += a; left -= a
            elif roll < opts["indel"] + opts["eqx"] + opts["pad"] and left > 2:
                a = int(rng.integers(1, left))
                ops.append(("M", a)); seq.append(G[r:r + a]); r += a; left -= a
                ops.append(("P", int(rng.integers(1, 3))))
            else:
                ops.append(("M", left)); seq.append(G[r:r + left]); r += left; left = 0
    # merge adjacent identical ops
    merged = []
    for o_, l_ in ops:
        if merged and merged[-1][0] == o_ and o_ in "M=X":
            merged[-1][1] += l_
        else:
            merged.append([o_, l_])
    seq = "".join(seq)
    # substitutions
    if opts["sub"] > 0 and len(seq):
        arr = list(seq)
        for i in np.nonzero(rng.random(len(arr)) < opts["sub"])[0]:
            arr[i] = BASES[(BASES.find(arr[i]) + 1) % 4] if arr[i] in BASES else "A"
        seq = "".join(arr)
    # non-ACGT letters of the genome that are not in the nt16 alphabet cannot occur (genome is IUPAC only)
    # clips
    if rng.random() < opts["clip"]:
        n = int(rng.integers(1, 6)); merged.insert(0, ["S", n]); seq = "".join(rng.choice(list(BASES), size=n)) + seq
    if rng.random() < opts["clip"]:
        n = int(rng.integers(1, 6)); merged.append(["S", n]); seq = seq + "".join(rng.choice(list(BASES), size=n))
    if rng.random() < opts["hard"]:
        merged.insert(0, ["H", int(rng.integers(1, 9))])
    if rng.random() < opts["hard"]:
        merged.append(["H", int(rng.integers(1, 9))])
    cigar = "".join(f"{l}{o}" for o, l in merged)
    return dict(pos=pos, cigar=cigar, seq=seq)


DEFAULT_OPTS = dict(indel=0.06, eqx=0.04, pad=0.01, sub=0.01, clip=0.1, hard=0.03)
# The "exception channel" family (round 6: k1_emit compares in 2 bits where reads and genome are pure ACGT): a CLEAN genome
# (genome_n = genome_iupac = 0) with characters outside ACGT -- N, IUPAC codes, lower-case letters -- planted at the anchors' edges
# (the first / last base of an exon, the bases either side of it, the boundaries of the exception bitmap's 64-base stretches), and
# reads that carry N / IUPAC codes / '=' of their own, at their anchors' first and last bases among others.  edge_exc plants before
# the reads are drawn (they copy the letters), late_exc afterwards (a pure-ACGT read over a letter that is not).
EXC_OPTS = dict(genome_n=0.0, genome_iupac=0.0, genome_lower=0.02, edge_exc=0.04, late_exc=0.08, read_exc=0.08)
_NON_ACGT = list("NNNNRYSWKMBDHVacgtn")


def plant_edge_exceptions(rng, genome, txs, p):
    g = list(genome)
    for ex in txs:
        for s, e in ex:
            for at in (s - 1, s, s + 1, e - 2, e - 1, e, (s // 64) * 64 + 63, (s // 64) * 64 + 64, (e // 64) * 64 - 1, (e // 64) * 64):
                if 0 <= at < len(g) and rng.random() < p / 3:
                    g[at] = str(rng.choice(_NON_ACGT))
    return "".join(g)


def read_exceptions(rng, r, p):
    """With probability p the read gets one to three letters outside ACGT ('=' among them): at the first / last base of one of
    its M blocks, or anywhere."""
    import re
    if r["seq"] is None or not r["seq"] or rng.random() >= p:
        return
    ops = [(int(n), o) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", r["cigar"])]
    edges = []
    q = 0
    for n, o in ops:
        if o in "M=X":
            edges += [q, q + n - 1]
        if o in "MIS=X":
            q += n
    arr = list(r["seq"])
    for _ in range(int(rng.integers(1, 4))):
        at = int(rng.choice(edges)) if edges and rng.random() < 0.6 else int(rng.integers(0, len(arr)))
        if 0 <= at < len(arr):
            arr[at] = str(rng.choice(list("NNNRYSWKMBDHV=")))
    r["seq"] = "".join(arr)


def make_reads(seed, glen=30000, n_reads=3000, paired=False, opts=None, L=(30, 150), n_tx=12, no_xs_frac=0.1,
               noseq_frac=0.01):
    rng = np.random.default_rng(seed)
    o = dict(DEFAULT_OPTS)
    if opts:
        o.update(opts)
    genome = random_genome(rng, glen, lower_frac=o.get("genome_lower", 0.1), n_frac=o.get("genome_n", 0.002), iupac_frac=o.get("genome_iupac", 0.001))
    txs, genome = make_transcripts(rng, glen, n_tx=n_tx, genome=genome)
    if o.get("edge_exc"):
        genome = plant_edge_exceptions(rng, genome, txs, o["edge_exc"])
    strands = [("+" if rng.random() < 0.5 else "-") for _ in txs]
    # zipf-ish transcript popularity
    w = 1.0 / np.arange(1, len(txs) + 1) ** 1.1
    w /= w.sum()
    reads = []
    for _ in range(n_reads):
        t = int(rng.choice(len(txs), p=w))
        r = _tx_read(rng, genome, txs[t], int(rng.integers(L[0], L[1] + 1)), o)
        if r is None:
            continue
        rev = rng.random() < 0.5
        if paired:
            first = rng.random() < 0.5
            flag = 1 | (0x40 if first else 0x80) | (0x10 if rev else 0x20)
            if rng.random() < 0.8:
                flag |= 2
            if rng.random() < 0.05:
                flag |= 8
            r["mtid"] = 0 if rng.random() < 0.95 else 1
            r["mpos"] = max(0, r["pos"] + int(rng.integers(-400, 400)))
        else:
            flag = 0x10 if rev else 0
        if rng.random() < 0.03:
            flag |= 0x100
        if rng.random() < 0.03:
            flag |= 0x400
        r["flag"] = flag
        r["mapq"] = int(rng.choice([60, 60, 60, 60, 30, 29, 3, 0]))
        x = rng.random()
        r["xs"] = None if x < no_xs_frac else (strands[t] if x < 0.97 else ("-" if strands[t] == "+" else "+"))
        if o.get("read_exc"):
            read_exceptions(rng, r, o["read_exc"])
        if rng.random() < noseq_frac:
            r["seq"] = None
        reads.append(r)
    if o.get("late_exc"):
        genome = plant_edge_exceptions(rng, genome, txs, o["late_exc"])
    reads.sort(key=lambda r: r["pos"])
    return genome, reads


def to_batch(reads):
    return ReadBatch.from_reads(reads)
