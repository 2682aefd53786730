"""Target groups (pjb_finish_group_begin / _end): several targets finished as ONE kernel chain must give exactly the rows
and per-target results of finishing them one by one -- and those of the oracle."""
import numpy as np
import pytest

from fuzzgen import make_reads, to_batch
from parity import assert_rows_equal, region_equal
from portcullis_amd.records import ReadBatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def ffi():
    from portcullis_amd import ffi
    assert ffi.device_count() >= 1, "no HIP device visible"
    return ffi


def _contigs(orc, seeds, orientation="FR", n_reads=2500, paired=True):
    out = []
    for tid, seed in enumerate(seeds):
        genome, reads = make_reads(seed, n_reads=n_reads, paired=paired)
        batch = to_batch(reads)
        orows, oreg = orc.find_juncs(tid, len(genome), genome, batch, orientation)
        out.append((genome, batch, orows, oreg))
    return out


def _setup(ctx, contigs):
    ctx.set_refs([len(g) for g, _, _, _ in contigs])
    for tid, (g, _, _, _) in enumerate(contigs):
        ctx.upload_contig(tid, g.encode() if isinstance(g, str) else g)
    ctx.clear_rows()


def _singles(ctx, contigs):
    _setup(ctx, contigs)
    regs = {}
    for tid, (_, b, _, _) in enumerate(contigs):
        ctx.submit_batch(tid, b)
        regs[tid] = ctx.finish_contig(tid)
    return ctx.collect(), regs


def _grouped(ctx, contigs, groups, split=None):
    """groups: list of tid lists, all begun before the first is collected (up to MAX_QUEUED)."""
    _setup(ctx, contigs)
    for tid, (_, b, _, _) in enumerate(contigs):
        if b is None:
            continue
        if split and b.n > 10:
            cut = b.n // 3
            ctx.submit_batch(tid, b.slice(0, cut))
            ctx.submit_batch(tid, b.slice(cut, b.n))
        else:
            ctx.submit_batch(tid, b)
    for g in groups:
        ctx.finish_group_begin(g)
    regs = {}
    for g in groups:
        regs.update(ctx.finish_group_end(g))
    return ctx.collect(), regs


@pytest.mark.parametrize("orientation", ["FR", "UNKNOWN"])
def test_group_equals_singles_and_oracle(ffi, orc, orientation):
    contigs = _contigs(orc, (41, 42, 43, 44, 45), orientation)
    want = np.concatenate([c[2] for c in contigs])
    with ffi.Context(0, orientation) as ctx:
        srows, sregs = _singles(ctx, contigs)
        assert_rows_equal(srows, want)
        for groups, split in (([[0, 1, 2, 3, 4]], False), ([[0, 1], [2, 3, 4]], True), ([[0], [1, 2, 3], [4]], False)):
            rows, regs = _grouped(ctx, contigs, groups, split)
            assert rows.tobytes() == srows.tobytes(), groups
            for tid, c in enumerate(contigs):
                region_equal(regs[tid], c[3])
                assert regs[tid] == sregs[tid], (tid, regs[tid], sregs[tid])


def test_group_with_empty_targets_and_order(ffi, orc):
    """Targets without alignments may be named; the rows come in the order of `tids`, whatever that order is."""
    four = []
    for tid, seed in enumerate((51, None, 52, 53)):  # (per-read predicates compare the mate's target with the read's own: the oracle gets the same indices)
        if seed is None:
            four.append(("ACGT" * 500, None, None, None))
            continue
        genome, reads = make_reads(seed, n_reads=2500, paired=True)
        batch = to_batch(reads)
        orows, oreg = orc.find_juncs(tid, len(genome), genome, batch, "FR")
        four.append((genome, batch, orows, oreg))
    with ffi.Context(0, "FR") as ctx:
        rows, regs = _grouped(ctx, four, [[0, 1, 2, 3]])
        assert regs[1]["n_reads"] == 0 and regs[1]["n_junctions"] == 0 and regs[1]["min_len"] == 2**31 - 1
        for tid in (0, 2, 3):
            region_equal(regs[tid], four[tid][3])
        assert_rows_equal(rows[rows["refid"] == 0], four[0][2])
        assert_rows_equal(rows[rows["refid"] == 2], four[2][2])
        assert_rows_equal(rows[rows["refid"] == 3], four[3][2])
        # another order: rows follow it
        rows2, _ = _grouped(ctx, four, [[3, 0, 2]])
        assert list(dict.fromkeys(rows2["refid"].tolist())) == [3, 0, 2]
        assert_rows_equal(rows2[rows2["refid"] == 3], four[3][2])
        assert_rows_equal(rows2[rows2["refid"] == 0], four[0][2])
        # a group of targets that all lack alignments
        ctx.set_refs([1000, 2000])
        ctx.clear_rows()
        ctx.finish_group_begin([0, 1])
        r = ctx.finish_group_end([0, 1])
        assert r[0]["n_reads"] == 0 and r[1]["n_reads"] == 0 and len(ctx.collect()) == 0


def test_group_member_with_alignments_outside_its_sequence(ffi, orc):
    """An alignment that runs past the end of its own target must not see the next member's bases: the group is taken apart
    and its members are finished one by one -- which reports exactly what finishing that target alone reports (here the
    reference's "anchor region ... not the same size" condition, an error in the oracle as well)."""
    contigs = _contigs(orc, (61, 62), "UNKNOWN", paired=False)
    from fixtures_micro import read_from_genome
    rng = np.random.default_rng(99)
    genome = "".join(rng.choice(list("ACGT"), size=6000))
    glen = len(genome)
    odd = ReadBatch.from_reads([read_from_genome(genome, 5900, "50M100N60M")])  # (test_gpu_edge_cases: read_runs_off_contig_end)
    with pytest.raises(orc.OracleError):
        orc.find_juncs(2, glen, genome, odd, "UNKNOWN")
    three = [contigs[0], contigs[1], (genome, odd, None, None)]
    with ffi.Context(0, "UNKNOWN") as ctx:
        _setup(ctx, three)
        ctx.submit_batch(2, odd)
        with pytest.raises(ffi.PjbError) as single:
            ctx.finish_contig(2)
    with ffi.Context(0, "UNKNOWN") as ctx:
        _setup(ctx, three)
        for tid in range(3):
            ctx.submit_batch(tid, three[tid][1])
        ctx.finish_group_begin([0, 2, 1])
        with pytest.raises(ffi.PjbError) as grouped:
            ctx.finish_group_end([0, 2, 1])
        assert grouped.value.code == single.value.code, (grouped.value, single.value)
        # the context goes on: the two healthy targets as a group
        ctx.clear_rows()
        ctx.submit_batch(0, three[0][1])
        ctx.submit_batch(1, three[1][1])
        ctx.finish_group_begin([0, 1])
        regs = ctx.finish_group_end([0, 1])
        region_equal(regs[0], three[0][3])
        assert_rows_equal(ctx.collect(), np.concatenate([three[0][2], three[1][2]]))


def test_group_refusals(ffi, orc):
    contigs = _contigs(orc, (71, 72))
    with ffi.Context(0, "FR") as ctx:
        _setup(ctx, contigs)
        with pytest.raises(ffi.PjbError):
            ctx.finish_group_begin([])
        with pytest.raises(ffi.PjbError):
            ctx.finish_group_begin([0, 0])
        with pytest.raises(ffi.PjbError):
            ctx.finish_group_begin([0, 7])
        ctx.submit_batch(0, contigs[0][1])
        ctx.submit_batch(1, contigs[1][1])
        ctx.finish_group_begin([0, 1])
        with pytest.raises(ffi.PjbError):  # collected with other targets than it was begun with: refused, the chain stays queued
            ctx.finish_group_end([1, 0])
        with pytest.raises(ffi.PjbError):
            ctx.finish_contig_end(0)
        regs = ctx.finish_group_end([0, 1])
        region_equal(regs[0], contigs[0][3])
        assert_rows_equal(ctx.collect(), np.concatenate([contigs[0][2], contigs[1][2]]))
    # a genome with characters outside the 16-letter alphabet cannot be a member
    g = contigs[0][0][:200] + "J" + contigs[0][0][201:]
    with ffi.Context(0, "FR") as ctx:
        ctx.set_refs([len(g), len(contigs[1][0])])
        ctx.upload_contig(0, g.encode())
        ctx.upload_contig(1, contigs[1][0].encode())
        ctx.submit_batch(0, contigs[0][1])
        ctx.submit_batch(1, contigs[1][1])
        with pytest.raises(ffi.PjbError):
            ctx.finish_group_begin([0, 1])
        ctx.finish_contig(0)
        ctx.finish_contig(1)


# ---- the limits (include/portcullis_amd.h: PJB_GROUP_MAX members, a virtual sequence below 2^31 bases, pair limit of a chain)
def test_group_of_the_maximum_number_of_members(ffi, orc):
    """PJB_GROUP_MAX (32) targets in one chain -- some of them without a single alignment -- and one more is refused."""
    seeds = list(range(300, 300 + ffi.GROUP_MAX + 1))
    contigs = _contigs(orc, seeds, n_reads=300)
    for k in (3, 17, ffi.GROUP_MAX - 1):  # members without reads
        g, b, orows, _ = contigs[k]
        contigs[k] = (g, None, orows[:0], None)
    with ffi.Context(0, "FR") as ctx:
        _setup(ctx, contigs)
        for tid, (_, b, _, _) in enumerate(contigs):
            if b is not None:
                ctx.submit_batch(tid, b)
        with pytest.raises(ffi.PjbError):
            ctx.finish_group_begin(list(range(ffi.GROUP_MAX + 1)))
        members = list(range(ffi.GROUP_MAX))
        ctx.finish_group_begin(members)
        regs = ctx.finish_group_end(members)
        last = ctx.finish_contig(ffi.GROUP_MAX)
        rows = ctx.collect()
    want = np.concatenate([c[2] for c in contigs])
    assert_rows_equal(rows, want)
    for tid in members:
        if contigs[tid][3] is not None:
            region_equal(regs[tid], contigs[tid][3])
        else:
            assert regs[tid]["n_reads"] == 0 and regs[tid]["n_junctions"] == 0
    region_equal(last, contigs[ffi.GROUP_MAX][3])


def test_group_just_under_two_to_the_31_bases(ffi, orc):
    """A group's virtual sequence must stay below 2^31 bases: three targets declared 720 Mb long (their alignments sit in the first
    30 kb; the genome a chain reads is the uploaded one, so the upload is a device buffer of zeros with the real bases in front)
    make a group of 2.16 G (2^31 = 2.147 G) -- refused -- while two of them plus a small one (1.44 G: virtual offsets far beyond 2^30) work."""
    import torch
    contigs = _contigs(orc, (411, 412, 413, 414), n_reads=1500)
    big = 720_000_000
    lens = [big, big, big, len(contigs[3][0])]
    with ffi.Context(0, "FR") as ctx:
        ctx.set_refs(lens)
        keep = []
        for tid in range(3):
            g = contigs[tid][0].upper().encode()   # (pjb_upload_contig_device takes upper-cased bases)
            d = torch.full((big,), ord("N"), dtype=torch.uint8, device="cuda")
            d[: len(g)] = torch.frombuffer(bytearray(g), dtype=torch.uint8).cuda()
            ctx.upload_contig_device(tid, d)
            keep.append(d)
        ctx.upload_contig(3, contigs[3][0].encode())
        # the oracle saw targets of 30 kb; rows depend on the target's length only through clamps at its end, which these reads never reach
        ctx.clear_rows()
        for tid in range(4):
            ctx.submit_batch(tid, contigs[tid][1])
        with pytest.raises(ffi.PjbError):
            ctx.finish_group_begin([0, 1, 2])          # 2.16 G bases
        ctx.finish_group_begin([0, 1, 3])              # 1.44 G: member 3 starts beyond 2^30
        regs = ctx.finish_group_end([0, 1, 3])
        r2 = ctx.finish_contig(2)
        rows = ctx.collect()
        for tid in (0, 1, 3):
            region_equal(regs[tid], contigs[tid][3])
        region_equal(r2, contigs[2][3])
        want = np.concatenate([contigs[t][2] for t in (0, 1, 3, 2)])
        assert_rows_equal(rows, want)
        del keep


def test_pair_limit_overflow_inside_a_group(ffi, orc):
    """A chain is queued with room for 5/8 pairs per alignment (+ 4096); reads with many introns each exceed that: the control block
    reports the overflow, the group is repeated once with the exact count -- and the chain queued behind it with it."""
    def many_intron_reads(seed, n):
        rng = np.random.default_rng(seed)
        genome = "".join(rng.choice(list("ACGT"), size=60000))
        reads = []
        for k in range(n):
            pos = int(rng.integers(0, 2000))
            segs = int(rng.integers(6, 10))
            cigar, seq, r = "", [], pos
            for s in range(segs):
                m = int(rng.integers(8, 20))
                cigar += f"{m}M"
                seq.append(genome[r:r + m])
                r += m
                if s + 1 < segs:
                    nl = int(rng.integers(50, 300))
                    cigar += f"{nl}N"
                    r += nl
            reads.append(dict(pos=pos, cigar=cigar, seq="".join(seq), flag=0, mapq=60, xs="+", mtid=-1, mpos=-1))
        reads.sort(key=lambda x: x["pos"])
        return genome, reads

    contigs = []
    for tid, (seed, n) in enumerate(((901, 9000), (902, 7000), (903, 2500))):
        genome, reads = many_intron_reads(seed, n)
        batch = to_batch(reads)
        orows, oreg = orc.find_juncs(tid, len(genome), genome, batch, "UNKNOWN")
        assert oreg["spliced"] == n and len(orows) > 100
        contigs.append((genome, batch, orows, oreg))
    with ffi.Context(0, "UNKNOWN") as ctx:
        _setup(ctx, contigs)
        for tid, (_, b, _, _) in enumerate(contigs):
            ctx.submit_batch(tid, b)
        ctx.finish_group_begin([0, 1])     # ~100 k pairs from 16 k alignments: over the limit
        ctx.finish_contig_begin(2)         # queued behind it
        regs = ctx.finish_group_end([0, 1])
        r2 = ctx.finish_contig_end(2)
        rows = ctx.collect()
    for tid in (0, 1):
        region_equal(regs[tid], contigs[tid][3])
        assert regs[tid]["n_pairs"] > 5 * regs[tid]["n_reads"]
    region_equal(r2, contigs[2][3])
    assert_rows_equal(rows, np.concatenate([c[2] for c in contigs]))


@pytest.mark.parametrize("dense", [1, 0])
def test_read_list_room_overflow_is_repeated(ffi, orc, dense):
    """nothing reached the OVF_LISTS path (a read sub-list that overflows closes the chain, which is queued again
    with the room it asked for).  pjb_set_option("list_cap", 8) makes every chain's first attempt overflow: groups == singles ==
    oracle after the repeat, on the dense-id chain and on the full-key chain."""
    contigs = _contigs(orc, [501, 502, 503], n_reads=3000)
    with ffi.Context(0, "FR") as ctx:
        ctx.set_option("dense_ids", dense)
        ctx.set_option("list_cap", 8)
        rows_s, regs_s = _singles(ctx, contigs)
        t = ctx.timing()
        assert t["repeats"] >= 1 and t["repeat_reasons"] & 16, t  # (the last single chain: queued again for the read lists' room)
        rows_g, regs_g = _grouped(ctx, contigs, [[0, 1, 2]])
        t = ctx.timing()
        assert t["repeats"] >= 1 and t["repeat_reasons"] & 16, t
    want = np.concatenate([c[2] for c in contigs])
    assert_rows_equal(rows_s, want)
    assert rows_g.tobytes() == rows_s.tobytes()
    for tid, c in enumerate(contigs):
        region_equal(regs_s[tid], c[3])
        region_equal(regs_g[tid], c[3])


def test_sort_digits_planned_too_small_are_repeated(ffi, orc):
    """The sort's digits are planned from the junctions per read the context's chains have had (twice that, + 64, at least
    "sort_floor"), not from the buffers' limit; a chain with more junctions than that closes (OVF_JUNC from kd_table) and is repeated
    with digits for the limit.  A first target with thousands of reads on a handful of junctions makes the plan small, the targets
    behind it hold ~90 junctions in under a thousand reads: their chains MUST repeat -- pjb_timing.repeats says so -- singly and as a
    group, and still give the oracle's rows (a sort with too few digit bits would misorder them)."""
    from fuzzgen import make_reads, to_batch
    fixed = []
    for tid, (seed, n_reads, n_tx) in enumerate([(611, 4000, 1), (612, 800, 40), (613, 900, 40)]):
        g, reads = make_reads(seed, glen=40000, n_reads=n_reads, paired=True, n_tx=n_tx)
        b = to_batch(reads)
        orows, oreg = orc.find_juncs(tid, len(g), g, b, "FR")
        fixed.append((g, b, orows, oreg))
    assert len(fixed[0][2]) < 10 and len(fixed[1][2]) > 80 and len(fixed[2][2]) > 80
    want = np.concatenate([c[2] for c in fixed])

    def run(groups):
        with ffi.Context(0, "FR") as ctx:
            ctx.set_option("sort_floor", 2)
            _setup(ctx, fixed)
            regs, repeats = {}, []
            for g in groups:
                for tid in g:
                    ctx.submit_batch(tid, fixed[tid][1])
                if len(g) == 1:
                    regs[g[0]] = ctx.finish_contig(g[0])
                else:
                    ctx.finish_group_begin(g)
                    regs.update(ctx.finish_group_end(g))
                t = ctx.timing()
                repeats.append((t["repeats"], t["repeat_reasons"]))
            return ctx.collect(), regs, repeats

    rows_s, regs_s, rep_s = run([[0], [1], [2]])
    # the second chain was queued again for its junctions (the first one may repeat for its read lists' room -- thousands of reads of one
    # transcript on a few sub-lists -- but not for junctions: nothing had been planned yet)
    assert not rep_s[0][1] & 4 and rep_s[1][0] >= 1 and rep_s[1][1] & 4, rep_s
    assert_rows_equal(rows_s, want)
    rows_g, regs_g, rep_g = run([[0], [1, 2]])
    assert rep_g[1][0] >= 1 and rep_g[1][1] & 4, rep_g
    assert rows_g.tobytes() == rows_s.tobytes()
    for tid, c in enumerate(fixed):
        region_equal(regs_s[tid], c[3])
        region_equal(regs_g[tid], c[3])
