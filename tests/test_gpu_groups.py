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
