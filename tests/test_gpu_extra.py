"""`junc --extra` on the device against the oracle: mm_score, coverage, up_aln, down_aln,
through the C ABI (PJB_FLAG_EXTRA contexts, pjb_extra_finish)."""
import numpy as np
import pytest

from extra_util import add_names, assert_extra_equal, device_extra, oracle_extra
from fuzzgen import make_reads
from parity import assert_rows_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ffi():
    from portcullis_amd import ffi as f
    assert f.device_count() >= 1
    return f


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    return o


def _contigs(seed, n_contigs=3, n_reads=2500, paired=False, **kw):
    rng = np.random.default_rng(seed)
    pool = []
    out = []
    for t in range(n_contigs):
        genome, reads = make_reads(seed * 10 + t, n_reads=n_reads, paired=paired, **kw)
        add_names(reads, rng, f"c{t}", pool)
        out.append((genome, reads))
    return out


@pytest.mark.parametrize("seed,paired,queue,dense", [(1, False, 1, False), (2, True, 3, False), (3, False, 2, True), (4, True, 1, True),
                                                      (5, False, 4, False), (6, True, 2, False)])
def test_extra_fuzz_multi_contig(ffi, orc, seed, paired, queue, dense):
    """mm_score, coverage, up_aln, down_aln of three targets against the oracle: one target at a time and with several
    chains queued; through the records themselves (the default) and through the depth vector (`dense`)."""
    contigs = _contigs(seed, paired=paired, n_contigs=3 if seed < 5 else 5)
    orows, _ = oracle_extra(orc, contigs, "FR" if paired else "UNKNOWN")
    rows, extra = device_extra(ffi, orc, contigs, "FR" if paired else "UNKNOWN", queue=queue, dense=dense)
    assert_rows_equal(rows, orows)
    assert_extra_equal(rows, extra, orows)
    assert (extra["up_aln"] > 0).any() and (extra["down_aln"] > 0).any() and (extra["mm_score"] < 1).any()


def test_extra_batches_and_empty_targets(ffi, orc):
    """Ragged batches; a target without any record, one with spliced records only (it never enters the pileup, so
    the depth hand-over skips it), and the last target with unspliced records only."""
    a = _contigs(7, n_contigs=2, n_reads=1800)
    g2, r2 = make_reads(71, n_reads=600)
    r2 = [r for r in r2 if "N" in r["cigar"]]
    g4, r4 = make_reads(72, n_reads=400)
    r4 = [r for r in r4 if "N" not in r["cigar"]]
    rng = np.random.default_rng(5)
    add_names(r2, rng, "x2")
    add_names(r4, rng, "x4")
    contigs = [a[0], ("ACGT" * 500, None), (g2, r2), a[1], (g4, r4)]
    orows, _ = oracle_extra(orc, contigs)
    rows, extra = device_extra(ffi, orc, contigs, split=(0.1, 0.55, 0.56))
    assert_rows_equal(rows, orows)
    assert_extra_equal(rows, extra, orows)
    assert (extra["coverage"][rows["refid"] == 0] == 0).all()      # first target: never visited
    assert (extra["coverage"][rows["refid"] == 2] == 0).all()      # spliced-only target: not in the pileup, never visited
    assert (extra["coverage"][rows["refid"] == 3] != 0).any()      # gets target 0's depth vector (the hand-over quirk)


def test_extra_pileup_cap(ffi, orc):
    """More than 8000 unspliced records buffered with ties in position: htslib's pileup drops records
    (sam.c:1906); the device replays the cap over the hot span."""
    genome, reads = make_reads(11, glen=6000, n_reads=1500, L=(60, 120))
    rng = np.random.default_rng(3)
    spliced = [r for r in reads if "N" in r["cigar"]]
    assert spliced
    anchor = spliced[len(spliced) // 2]["pos"]
    deep = []
    for k in range(12000):   # a pile of duplicates and near-duplicates around a junction
        p = max(0, anchor - 40 + int(rng.integers(0, 6)))
        deep.append(dict(pos=p, cigar=f"{int(rng.integers(40, 90))}M", seq=None, l_qseq=0, flag=0))
    allr = sorted(reads + deep, key=lambda r: r["pos"])
    add_names(allr, rng, "d", unmapped_frac=0.0)
    contigs = [(genome, allr)]
    orows, _ = oracle_extra(orc, contigs)
    depth, kept = orc.depth(len(genome), __import__("extra_util").batch_with_names(orc, allr))
    n_unspliced = sum(1 for r in allr if "N" not in r["cigar"] and not (r.get("flag", 0) & 4))
    assert kept < n_unspliced and depth.max() >= 7999          # the cap really dropped records in the oracle
    rows, extra = device_extra(ffi, orc, contigs)
    assert_extra_equal(rows, extra, orows)


def test_extra_zero_span_and_edges(ffi, orc):
    """Mapped records without a reference span (getEnd() == pos - 1), junctions at the contig edge (windows clipped by
    the bounds test of calcCoverage), SEQ '*'."""
    g = "ACGTTGCAAC" * 60
    reads = [dict(pos=0, cigar="10M20N30M", seq="A" * 40, name="e0", xs="+"),
             dict(pos=5, cigar="12S", seq="A" * 12, name="z0"),            # no reference span
             dict(pos=9, cigar="30M", seq="A" * 30, name="u0"),
             dict(pos=10, cigar="4I", seq="A" * 4, name="z1"),             # pos == intron start == ... edge of the tests
             dict(pos=30, cigar="8S", seq="A" * 8, name="z2"),
             dict(pos=31, cigar="10M", seq=None, l_qseq=10, name="u1"),
             dict(pos=540, cigar="30M20N10M", seq="A" * 40, name="e1", xs="-"),
             dict(pos=590, cigar="9M", seq="A" * 9, name="u2")]
    contigs = [(g, reads)]
    orows, _ = oracle_extra(orc, contigs)
    rows, extra = device_extra(ffi, orc, contigs)
    assert_rows_equal(rows, orows)
    assert_extra_equal(rows, extra, orows)


def test_extra_gaps_and_dense_fallbacks(ffi, orc):
    """Deletions inside unspliced records around a junction (no depth inside a D), and the two ways a target falls back to
    the depth vector: a record with more than 126 deletions, more deletions than the gap list holds."""
    genome, reads = make_reads(21, glen=8000, n_reads=900, L=(60, 120))
    rng = np.random.default_rng(8)
    spliced = [r for r in reads if "N" in r["cigar"]]
    assert spliced
    extra_reads = []
    for r in spliced[::3]:
        import re
        ops = re.findall(r"(\d+)([MIDNSHP=X])", r["cigar"])
        start = r["pos"]         # intron start: the donor windows are [start - 21, start - 1]
        for ln, op in ops:
            if op == "N":
                break
            if op in "MD=X":
                start += int(ln)
        for k in range(4):
            p = max(0, start - 30 - 3 * k)
            extra_reads.append(dict(pos=p, cigar=f"{8 + k}M{2 + k}D{12}M1D{9}M", seq=None, l_qseq=0, flag=0))
    base = sorted(reads + extra_reads, key=lambda r: r["pos"])
    variants = {"gaps": base}
    many = dict(pos=max(0, spliced[0]["pos"] - 50), cigar="".join("1M1D" for _ in range(130)) + "5M", seq=None, l_qseq=0, flag=0)
    variants["a record with 130 gaps"] = sorted(base + [many], key=lambda r: r["pos"])
    few = [r for r in base if "N" in r["cigar"]][:40]
    lots = [dict(pos=max(0, few[k % len(few)]["pos"] - 20), cigar="".join("2M1D" for _ in range(60)) + "3M", seq=None, l_qseq=0, flag=0) for k in range(40)]
    variants["gap list full"] = sorted(few + lots, key=lambda r: r["pos"])
    for what, rr in variants.items():
        rr = [dict(r) for r in rr]
        add_names(rr, rng, "g", unmapped_frac=0.0)
        contigs = [(genome, rr)]
        orows, _ = oracle_extra(orc, contigs)
        for dense in (False, True):
            rows, extra = device_extra(ffi, orc, contigs, dense=dense)
            assert_rows_equal(rows, orows)
            assert_extra_equal(rows, extra, orows)
        assert (extra["coverage"] != 0).any(), what


def test_extra_needs_name_hash_and_flag(ffi, orc):
    genome, reads = make_reads(5, n_reads=300)
    from fuzzgen import to_batch
    with ffi.Context(0, "UNKNOWN", flags=ffi.FLAG_EXTRA) as ctx:
        ctx.set_refs([len(genome)])
        ctx.upload_contig(0, genome.encode())
        with pytest.raises(ffi.PjbError):
            ctx.submit_batch(0, to_batch(reads))          # no name_hash
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([len(genome)])
        with pytest.raises(ffi.PjbError):
            ctx.extra_finish()                            # not an --extra context
