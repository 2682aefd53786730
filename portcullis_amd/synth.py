"""Deterministic synthetic `junc` workloads (SURVEY.md section 8d / BASELINE.json configs).

Produces, directly in the `pjb_batch` layout, the alignment records a prepared
BAM of the given shape would decode to, plus the contig.  Written with torch so
the 10 M-read configuration is generated in HBM in about a second (bench.py) and
the same code makes small CPU cases for the parity tests.

Shape of the data (single-end configs):
  * contig: uniform random ACGT
  * junctions: uniform positions, intron length ~ lognormal(ln 1500, 1.2) clipped
    to [40, 200 k]; 85 % GT..AG (half as CT..AC, XS '-'), 5 % GC..AG / AT..AC, 10 % none;
    ~15 % share a donor or an acceptor with the previous junction; ~12 % are
    followed by a short exon (20-60 bp) and a second junction, so reads span 2-3 junctions
  * reads: length L; 30 % spliced, junction drawn from Zipf(1.1); left overhang U[1, L-1];
    ~3 % soft clips, ~2 % 1-3 bp insertion/deletion; 0.5 % substitutions; mapq 60/3/0
    = 90/7/3 %; XS:A on spliced reads; 5 % exact duplicates; coordinate sorted
"""
import math
from dataclasses import dataclass

import torch

OP_M, OP_I, OP_D, OP_N, OP_S = 0, 1, 2, 3, 4


@dataclass
class SynthConfig:
    name: str = "C2"
    contig_len: int = 100_000_000
    n_reads: int = 10_000_000
    n_junctions: int = 50_000
    read_len: int = 100
    spliced_frac: float = 0.30
    zipf_s: float = 1.1
    seed: int = 20260101
    paired: bool = False   # FR paired-end flags / mate fields (BASELINE configs[2..4])


CONFIGS = {
    # BASELINE.json configs[1]: synthetic 10M single-end reads, 1 contig, ~50k junctions
    "C2": SynthConfig(),
    "C2-small": SynthConfig("C2-small", 2_000_000, 200_000, 1_000, 100),
    "C2-tiny": SynthConfig("C2-tiny", 200_000, 20_000, 120, 100),
    # scaled-down shape of BASELINE configs[2]: paired-end 150-bp reads; used per contig by the multi-contig tests
    "C3-contig": SynthConfig("C3-contig", 1_000_000, 60_000, 400, 150, paired=True),
}


# BASELINE.json configs[2] / configs[3]: 200 M paired-end 150-bp reads over 25 contigs with the lengths of GRCh38
# chr1..22, X, Y, M (3.09 Gb); reads and ~250 k junctions spread in proportion to contig length.  One SynthConfig per
# contig (its own seed), so any rank can generate exactly the contigs it owns and get the same records as a
# single-GPU run of the whole set.
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
          135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
          46709983, 50818468, 156040895, 57227415, 16569]
GRCH38_NAMES = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY", "chrM"]


def c3_contig_configs(total_reads=200_000_000, total_junctions=250_000, read_len=150, lens=None, paired=True):
    """The per-contig configurations of the BASELINE configs[2] workload (scaled by the two totals)."""
    lens = GRCH38 if lens is None else lens
    tot = sum(lens)
    return [SynthConfig(f"C3-{i}", ln, max(200, round(total_reads * ln / tot)), max(2, round(total_junctions * ln / tot)),
                        read_len, paired=paired, seed=77_000 + i) for i, ln in enumerate(lens)]


def _randint(g, lo, hi, shape, dev, dtype=torch.int64):
    return torch.randint(int(lo), int(hi), shape, generator=g, device=dev, dtype=dtype)


def _rand(g, n, dev):
    return torch.rand(n, generator=g, device=dev)


def generate(cfg: SynthConfig, device="cpu", seed=None, tid=0):
    """Returns dict(genome=uint8[Lg] upper-case ASCII, batch={name: tensor}, n_reads, n_pairs, n_cigar_ops, ...)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(cfg.seed if seed is None else seed)
    Lg, N, J, L = cfg.contig_len, cfg.n_reads, cfg.n_junctions, cfg.read_len
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    genome = acgt[_randint(g, 0, 4, (Lg,), dev)]

    # ------------------------------------------------------------------ junctions
    margin = 4 * L + 64
    max_intron = min(200_000, Lg // 20)
    jstart = torch.sort(_randint(g, margin, Lg - 3 * max_intron - margin, (J,), dev)).values
    ilen = torch.exp(torch.randn(J, generator=g, device=dev) * 1.2 + math.log(1500.0)).clamp(40, max_intron).long()
    ids = torch.arange(J, device=dev)
    odd = (ids % 2 == 1)
    r = _rand(g, J, dev)
    share_d = odd & (r < 0.075)               # same donor as previous junction, different acceptor
    share_a = odd & (r >= 0.075) & (r < 0.15) # same acceptor as previous junction
    chain = odd & (r >= 0.15) & (r < 0.39)    # short exon after the previous junction (prev becomes a chain head)
    prev = (ids - 1).clamp(min=0)
    jend = jstart + ilen - 1
    e_len = _randint(g, 20, 61, (J,), dev)
    jstart = torch.where(share_d, jstart[prev], jstart)
    ilen = torch.where(share_d & (ilen == ilen[prev]), ilen + 7, ilen)
    jend = jstart + ilen - 1
    jend = torch.where(share_a, jend[prev], jend)
    jstart = torch.where(share_a, (jend - ilen + 1).clamp(min=margin), jstart)
    jstart = torch.where(chain, jend[prev] + 1 + e_len, jstart)
    jend = torch.where(chain, jstart + ilen - 1, jend)
    nxt = torch.full((J,), -1, dtype=torch.int64, device=dev)
    nxt[prev[chain]] = ids[chain]
    # second-level chains: junction k (k % 4 == 2) follows k-1 after a short exon
    lvl2 = (ids % 4 == 2) & (_rand(g, J, dev) < 0.10) & (ids > 0)
    jstart = torch.where(lvl2, jend[prev] + 1 + e_len, jstart)
    jend = torch.where(lvl2, jstart + ilen - 1, jend)
    nxt[prev[lvl2]] = ids[lvl2]
    ilen = jend - jstart + 1
    ok = (jend < Lg - margin) & (ilen >= 1)
    nxt = torch.where(ok[nxt.clamp(min=0)] & (nxt >= 0), nxt, torch.full_like(nxt, -1))
    # motifs / strand
    m = _rand(g, J, dev)
    strand = torch.where(m < 0.425, 1, torch.where(m < 0.85, 2, torch.where(m < 0.90, 1, 0)))  # xs code; 0 = none
    strand = torch.where(strand == 0, _randint(g, 1, 3, (J,), dev), strand)
    donor = torch.tensor([list(b"GT"), list(b"CT"), list(b"GC"), list(b"AT")], dtype=torch.uint8, device=dev)
    accpt = torch.tensor([list(b"AG"), list(b"AC"), list(b"AG"), list(b"AC")], dtype=torch.uint8, device=dev)
    kind = torch.where(m < 0.425, 0, torch.where(m < 0.85, 1, torch.where(m < 0.875, 2, torch.where(m < 0.90, 3, -1))))
    pl = (kind >= 0) & ok
    kk = kind.clamp(min=0)
    def plant(idx, val):
        # junctions that share a donor or an acceptor write the same bases: the junction with the highest id wins,
        # whatever order the device processes an index_put with duplicate indices in (a rank that regenerates a
        # contig must get the bytes every other rank got)
        o = torch.argsort(idx, stable=True)
        i_s, v_s = idx[o], val[o]
        last = torch.ones_like(i_s, dtype=torch.bool)
        last[:-1] = i_s[1:] != i_s[:-1]
        genome[i_s[last]] = v_s[last]

    for t in range(2):
        plant((jstart + t)[pl], donor[kk[pl], t])
        plant((jend - 1 + t)[pl], accpt[kk[pl], t])

    # ------------------------------------------------------------------ reads
    is_spl = _rand(g, N, dev) < cfg.spliced_frac
    S = int(is_spl.sum())
    U = N - S
    # --- unspliced
    upos = _randint(g, 0, Lg - L - 16, (U,), dev)
    uv = _rand(g, U, dev)
    u_clip = uv < 0.03
    u_ins = (uv >= 0.03) & (uv < 0.04)
    u_del = (uv >= 0.04) & (uv < 0.05)
    u_x = _randint(g, 1, 6, (U,), dev)       # clip / indel size helper
    u_a = _randint(g, 10, L - 10, (U,), dev) # split point
    # slots: [S][M][I/D][M]
    u_len = torch.zeros((U, 4), dtype=torch.int64, device=dev)
    u_op = torch.tensor([OP_S, OP_M, OP_I, OP_M], device=dev).repeat(U, 1)
    u_len[:, 1] = L
    u_len[:, 0] = torch.where(u_clip, u_x, 0)
    u_len[:, 1] = torch.where(u_clip, L - u_x, u_len[:, 1])
    indel = u_ins | u_del
    y = u_x.clamp(max=3)
    u_len[:, 1] = torch.where(indel, u_a, u_len[:, 1])
    u_len[:, 2] = torch.where(indel, y, 0)
    u_len[:, 3] = torch.where(u_ins, L - u_a - y, torch.where(u_del, L - u_a, 0))
    u_op[:, 2] = torch.where(u_del, OP_D, OP_I)
    u_lq = torch.full((U,), L, dtype=torch.int64, device=dev)

    # --- spliced
    w = torch.arange(1, J + 1, device=dev, dtype=torch.float64).pow(-cfg.zipf_s)
    w = torch.where(ok[torch.arange(J, device=dev)], w, torch.zeros_like(w))
    perm = torch.randperm(J, generator=g, device=dev)
    wp = torch.zeros_like(w)
    wp[perm] = w  # weight of junction perm[r] is rank-r weight
    wp = torch.where(ok, wp, torch.zeros_like(wp))
    cdf = torch.cumsum(wp, 0)
    cdf = cdf / cdf[-1]
    jx = torch.searchsorted(cdf, torch.rand(S, generator=g, device=dev, dtype=torch.float64)).clamp(max=J - 1)
    a = _randint(g, 1, L, (S,), dev)
    var = _rand(g, S, dev)
    sflag = _randint(g, 0, 2, (S,), dev) * 16
    mq = _rand(g, S, dev)
    smapq = torch.where(mq < 0.90, 60, torch.where(mq < 0.97, 3, 0))
    sx = _randint(g, 1, 6, (S,), dev)
    # duplicates: copy the defining draws of another spliced read
    dup = _rand(g, S, dev) < 0.05
    src = _randint(g, 0, max(S, 1), (S,), dev)
    src = torch.where(dup, src, torch.arange(S, device=dev))
    jx, a, var, sflag, smapq, sx = jx[src], a[src], var[src], sflag[src], smapq[src], sx[src]
    rem = L - a
    n1 = ilen[jx]
    nx1 = nxt[jx]
    e1 = jstart[nx1.clamp(min=0)] - jend[jx] - 1
    c1 = (nx1 >= 0) & (rem > e1) & (e1 > 0)
    m1 = torch.where(c1, e1, rem)
    rem1 = rem - m1
    n2 = torch.where(c1, ilen[nx1.clamp(min=0)], 0)
    nx2 = torch.where(c1, nxt[nx1.clamp(min=0)], torch.full_like(nx1, -1))
    e2 = jstart[nx2.clamp(min=0)] - jend[nx1.clamp(min=0)] - 1
    c2 = c1 & (nx2 >= 0) & (rem1 > e2) & (e2 > 0)
    m2 = torch.where(c1, torch.where(c2, e2, rem1), 0)
    rem2 = rem1 - m2
    n3 = torch.where(c2, ilen[nx2.clamp(min=0)], 0)
    m3 = torch.where(c2, rem2, 0)
    # slots: 0 S | 1 M0a | 2 I/D | 3 M0b | 4 N1 | 5 M1 | 6 N2 | 7 M2 | 8 N3 | 9 M3 | 10 S
    s_len = torch.zeros((S, 11), dtype=torch.int64, device=dev)
    s_op = torch.tensor([OP_S, OP_M, OP_I, OP_M, OP_N, OP_M, OP_N, OP_M, OP_N, OP_M, OP_S], device=dev).repeat(S, 1)
    s_len[:, 1] = a
    s_len[:, 4] = n1
    s_len[:, 5] = m1
    s_len[:, 6] = n2
    s_len[:, 7] = m2
    s_len[:, 8] = n3
    s_len[:, 9] = m3
    spos = jstart[jx] - a
    clipL = (var < 0.015) & (a > sx + 1)
    clipR = (var >= 0.015) & (var < 0.03)
    last = torch.where(c2, 9, torch.where(c1, 7, 5))
    last_len = s_len.gather(1, last[:, None])[:, 0]
    clipR = clipR & (last_len > sx + 1)
    ins = (var >= 0.03) & (var < 0.04) & (a >= 16)
    dele = (var >= 0.04) & (var < 0.05) & (a >= 16)
    y = sx.clamp(max=3)
    # left clip: first sx aligned bases become soft clip; pos moves right
    s_len[:, 0] = torch.where(clipL, sx, 0)
    s_len[:, 1] = torch.where(clipL, a - sx, s_len[:, 1])
    spos = torch.where(clipL, spos + sx, spos)
    # right clip
    s_len.scatter_(1, last[:, None], torch.where(clipR, last_len - sx, last_len)[:, None])
    s_len[:, 10] = torch.where(clipR, sx, 0)
    # indel inside the left anchor at offset 6 from the read start
    s_len[:, 1] = torch.where(ins | dele, 6, s_len[:, 1])
    s_len[:, 2] = torch.where(ins | dele, y, 0)
    s_op[:, 2] = torch.where(dele, OP_D, OP_I)
    # insertion keeps read length L: left anchor shrinks by y; deletion keeps aligned bases: anchor spans y more ref
    s_len[:, 3] = torch.where(ins, a - 6 - y, torch.where(dele, a - 6, 0))
    spos = torch.where(dele, spos - y, torch.where(ins, spos + y, spos))
    s_xs = strand[jx]
    s_lq = torch.full((S,), L, dtype=torch.int64, device=dev)

    # --- read bases for spliced reads: [S, L] genome coordinates (or -1 = random base)
    t = torch.arange(L, device=dev)[None, :]
    gidx = torch.full((S, L), -1, dtype=torch.int64, device=dev)
    qoff = torch.zeros(S, dtype=torch.int64, device=dev)
    rpos = spos.clone()
    for k in range(11):
        ln = s_len[:, k]
        op = s_op[:, k]
        cq = (op == OP_M) | (op == OP_I) | (op == OP_S)
        cr = (op == OP_M) | (op == OP_D) | (op == OP_N)
        isM = op == OP_M
        inside = (t >= qoff[:, None]) & (t < (qoff + ln)[:, None]) & (cq & (ln > 0))[:, None]
        val = torch.where(isM[:, None], rpos[:, None] + (t - qoff[:, None]), torch.full_like(gidx, -1))
        gidx = torch.where(inside, val, gidx)
        qoff = qoff + torch.where(cq, ln, torch.zeros_like(ln))
        rpos = rpos + torch.where(cr, ln, torch.zeros_like(ln))
    base = genome[gidx.clamp(min=0)]
    rnd = acgt[_randint(g, 0, 4, (S, L), dev)]
    base = torch.where(gidx >= 0, base, rnd)
    sub = torch.rand((S, L), generator=g, device=dev) < 0.005
    # substitute with a different base: rotate within ACGT
    code = torch.zeros(256, dtype=torch.uint8, device=dev)
    code[acgt.long()] = torch.arange(4, dtype=torch.uint8, device=dev)
    b4 = code[base.long()]
    b4 = torch.where(sub, (b4 + 1 + _randint(g, 0, 3, (S, L), dev, torch.uint8)) % 4, b4)
    nt16 = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=dev)[b4.long()]
    if L % 2:
        nt16 = torch.cat([nt16, torch.zeros((S, 1), dtype=torch.uint8, device=dev)], 1)
    packed = (nt16[:, 0::2] << 4) | nt16[:, 1::2]
    nbytes = packed.shape[1]
    W = (nbytes + 3) // 4
    if W * 4 != nbytes:
        packed = torch.cat([packed, torch.zeros((S, W * 4 - nbytes), dtype=torch.uint8, device=dev)], 1)

    # ------------------------------------------------------------------ merge + sort
    pos = torch.empty(N, dtype=torch.int64, device=dev)
    pos[is_spl] = spos
    pos[~is_spl] = upos
    order = torch.argsort(pos, stable=True)
    inv_spl = torch.full((N,), -1, dtype=torch.int64, device=dev)
    inv_spl[is_spl] = torch.arange(S, device=dev)
    inv_uns = torch.full((N,), -1, dtype=torch.int64, device=dev)
    inv_uns[~is_spl] = torch.arange(U, device=dev)
    o_spl = is_spl[order]
    o_s = inv_spl[order][o_spl]   # spliced ids in sorted order
    o_u = inv_uns[order][~o_spl]
    # ops: per read count, then flattened in sorted order
    MAXS = 11
    len_all = torch.zeros((N, MAXS), dtype=torch.int64, device=dev)
    op_all = torch.zeros((N, MAXS), dtype=torch.int64, device=dev)
    len_all[o_spl] = s_len[o_s]
    op_all[o_spl] = s_op[o_s]
    tmp = torch.zeros((U, MAXS), dtype=torch.int64, device=dev)
    tmp[:, :4] = u_len
    len_all[~o_spl] = tmp[o_u]
    tmp[:, :4] = u_op
    op_all[~o_spl] = tmp[o_u]
    present = len_all > 0
    n_ops = present.sum(1)
    cig_off = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    cig_off[1:] = torch.cumsum(n_ops, 0)
    cigar = ((len_all << 4) | op_all)[present].to(torch.int32)  # bit pattern of BAM uint32 (len < 2^27)
    flag = torch.zeros(N, dtype=torch.int64, device=dev)
    flag[o_spl] = sflag[o_s]
    flag[~o_spl] = _randint(g, 0, 2, (U,), dev)[o_u] * 16
    mapq = torch.full((N,), 60, dtype=torch.int64, device=dev)
    mapq[o_spl] = smapq[o_s]
    xs = torch.zeros(N, dtype=torch.int64, device=dev)
    xs[o_spl] = s_xs[o_s]
    lq = torch.full((N,), L, dtype=torch.int64, device=dev)
    seq_words = torch.where(o_spl, W, 0)
    seq_off = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    seq_off[1:] = torch.cumsum(seq_words, 0)
    seq4 = packed[o_s].reshape(-1).contiguous()
    # the same bases in 2 bits and the reads' exception bitmap (pjb_batch.seq2 / .seq_exc, ABI 4: what a decoder writes beside seq4; every
    # generated base is one of ACGT and every spliced read carries its L bases)
    seq2 = pack_seq2_torch(seq4)
    seq_exc = bits_to_words(~o_spl)
    n_pairs = int(((op_all == OP_N) & present).sum())
    mtid_t = torch.full((N,), -1, dtype=torch.int64, device=dev)
    mpos_t = torch.full((N,), -1, dtype=torch.int64, device=dev)
    if cfg.paired:
        # FR library: forward read upstream of its reverse mate.  ~8 % of pairs are broken in assorted ways.
        rev = (flag & 16) != 0
        first_mate = _randint(g, 0, 2, (N,), dev) == 1
        pr = _rand(g, N, dev)
        gap = _randint(g, 50, 400, (N,), dev)
        sp = pos[order]
        mp = torch.where(rev, sp - gap, sp + gap).clamp(min=0)
        wrong_side = (pr >= 0.92) & (pr < 0.94)
        mp = torch.where(wrong_side, torch.where(rev, sp + gap, (sp - gap).clamp(min=0)), mp)
        same_strand = (pr >= 0.94) & (pr < 0.96)
        mate_unmapped = (pr >= 0.96) & (pr < 0.98)
        other_contig = pr >= 0.98
        mrev = torch.where(same_strand, rev, ~rev)
        flag = flag | 1 | torch.where(first_mate, 0x40, 0x80) | torch.where(mrev, 0x20, 0) | torch.where(pr < 0.92, 2, 0) \
            | torch.where(mate_unmapped, 8, 0)
        mtid_t = torch.where(other_contig, tid + 1, tid) + torch.zeros_like(sp)
        mpos_t = mp
    batch = dict(
        pos=pos[order].to(torch.int32).contiguous(),
        flag=flag.to(torch.int16),  # bit pattern of uint16
        mapq=mapq.to(torch.uint8),
        xs=xs.to(torch.uint8),
        l_qseq=lq.to(torch.int32),
        mtid=mtid_t.to(torch.int32),
        mpos=mpos_t.to(torch.int32),
        cig_off=cig_off.to(torch.int32),
        cigar=cigar,
        seq_off=seq_off.to(torch.int32),
        seq4=seq4,
        seq2=seq2,
        seq_exc=seq_exc,
    )
    out = dict(genome=genome, batch=batch, n_reads=N, n_pairs=n_pairs, n_cigar_ops=int(cig_off[-1]),
               n_spliced=S, seq_words_per_read=W, config=cfg)
    if dev.type == "cuda":
        # the temporaries go back to the driver now: the library allocates with hipMalloc, not through torch's cache (with
        # max_split_size_mb set -- tests/conftest.py, bench.py -- the cache holds them as whole blocks that nothing pins)
        del pos, order, len_all, op_all, present, flag, mapq, xs, lq, seq_words, seq_off, cig_off, cigar, seq4, seq2, seq_exc, mtid_t, mpos_t, tmp
        torch.cuda.empty_cache()
    return out


def pack_seq2_torch(seq4):
    """uint8 tensor of 4-bit packed bases (whole 4-byte words) -> int16 tensor, one 16-bit granule per word (records.pack_seq2's layout), padded
    to an even number of granules."""
    code = torch.zeros(16, dtype=torch.uint8, device=seq4.device)
    code[torch.tensor([1, 2, 4, 8], device=seq4.device)] = torch.tensor([0, 1, 2, 3], dtype=torch.uint8, device=seq4.device)
    b = torch.arange(256, device=seq4.device)
    v2 = (code[b >> 4] | (code[b & 15] << 2)).to(torch.uint8)
    n = seq4.numel()
    out = torch.zeros(((n // 4 + 1) // 2) * 4, dtype=torch.uint8, device=seq4.device)
    STEP = 1 << 27
    for lo in range(0, n, STEP):
        v = v2[seq4[lo:lo + STEP].long()]
        out[lo // 2:(lo + v.numel()) // 2] = v[0::2] | (v[1::2] << 4)
    return out.view(torch.int16)


def bits_to_words(mask):
    """bool tensor -> int32 tensor of its bits, bit r & 31 of word r >> 5 (the bit pattern of uint32)."""
    n = mask.numel()
    m = torch.zeros(((n + 31) // 32) * 32, dtype=torch.int64, device=mask.device)
    m[:n] = mask.long()
    w = (m.view(-1, 32) << torch.arange(32, device=mask.device)).sum(1)
    return torch.where(w >= (1 << 31), w - (1 << 32), w).to(torch.int32)


def batch_to_numpy(batch, lo=0, hi=None):
    """Host ReadBatch of records [lo, hi) of a generated batch (for the oracle / host submits)."""
    import numpy as np

    from .records import ReadBatch

    n = batch["pos"].numel()
    hi = n if hi is None else hi
    cig_off = batch["cig_off"][lo:hi + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    seq_off = batch["seq_off"][lo:hi + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    c0, c1 = int(cig_off[0]), int(cig_off[-1])
    s0, s1 = int(seq_off[0]), int(seq_off[-1])

    def f(name, dt):
        return batch[name][lo:hi].cpu().numpy().view(dt)

    return ReadBatch(
        pos=f("pos", np.int32), flag=f("flag", np.uint16),
        mapq=f("mapq", np.uint8), xs=f("xs", np.uint8), l_qseq=f("l_qseq", np.int32), mtid=f("mtid", np.int32),
        mpos=f("mpos", np.int32), cig_off=(cig_off - c0).astype(np.uint32),
        cigar=batch["cigar"][c0:c1].cpu().numpy().view(np.uint32),
        seq_off=(seq_off - s0).astype(np.uint32), seq4=batch["seq4"][4 * s0:4 * s1].cpu().numpy(),
    )
