"""Edge cases the reference's code paths treat specially (clamps at contig ends, clipping quirks,
zero-length operations, exotic genome characters, error conditions).  For every case the HIP path
and the oracle must either both fail or produce identical rows."""
import numpy as np
import pytest

from fixtures_micro import read_from_genome
from parity import assert_rows_equal, region_equal
from portcullis_amd.records import ReadBatch

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(99)
G = "".join(RNG.choice(list("ACGT"), size=6000))


def both(ffi, orc, genome, reads, orientation="UNKNOWN", ref_len=None):
    ref_len = ref_len or len(genome)
    b = ReadBatch.from_reads(reads)
    oerr = derr = None
    orows = oreg = None
    try:
        orows, oreg = orc.find_juncs(0, ref_len, genome, b, orientation)
    except orc.OracleError as e:
        oerr = e
    with ffi.Context(0, orientation) as ctx:
        ctx.set_refs([ref_len])
        try:
            drows, dreg = ffi.run_contig(ctx, 0, genome.encode(), [b])
        except ffi.PjbError as e:
            derr = e
    assert (oerr is None) == (derr is None), f"oracle error: {oerr}; device error: {derr}"
    if oerr is None:
        region_equal(dreg, oreg)
        assert_rows_equal(drows, orows)
        return "ok", orows
    return "error", (oerr.code, derr.code)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def ffi():
    from portcullis_amd import ffi
    return ffi


def rd(pos, cigar, **kw):
    return read_from_genome(G, pos, cigar, **kw)


CASES = {
    "hardclip_then_softclip": [rd(1000, "5H3S50M100N47M"), rd(1010, "40M100N57M3S4H")],
    "softclip_both_ends": [rd(1000, "7S50M100N50M9S"), rd(1001, "49M100N51M")],
    "insertion_adjacent_to_intron": [rd(1000, "50M2I100N48M"), rd(1000, "50M100N3I47M"), rd(1005, "45M100N50M")],
    "deletion_adjacent_to_intron": [rd(1000, "48M2D100N50M"), rd(1000, "50M100N2D48M")],
    "pad_and_eqx_ops": [rd(1000, "20=5X25M1P100N10=40X"), rd(1002, "48M100N50M")],
    "back_op_and_unknown_op": [dict(pos=1000, cigar=np.array([(50 << 4) | 0, (3 << 4) | 9, (100 << 4) | 3, (50 << 4) | 0], np.uint32),
                                    seq=G[1000:1050] + G[1150:1200], xs="+")],
    "zero_length_refskip": [rd(1000, "50M0N50M"), rd(1000, "50M100N50M")],
    "zero_length_match_in_anchor": [rd(1000, "50M100N0M50M")],
    "two_introns_back_to_back": [rd(1000, "50M100N200N50M"), rd(1000, "50M100N10M200N40M")],
    "one_base_anchors": [rd(1049, "1M100N99M"), rd(1000, "50M100N1M"), rd(1020, "30M100N30M")],
    "read_of_length_one_base_each_side": [rd(1049, "1M100N1M")],
    "seq_length_one": [dict(pos=1000, cigar="50M100N50M", seq="A", xs="+")],
    "intron_enclosed_in_other_window": [rd(900, "150M100N50M"), rd(950, "30M40N30M100N50M"), rd(990, "60M100N40M")],
    "isoform_overlapping_region_end": [rd(1000, "50M100N50M"), rd(1000, "50M100N20M300N30M"), rd(1100, "60M40N50M")],
    "intron_at_contig_start": [rd(3, "6M100N50M")],
    "anchor_reaches_contig_end": [rd(5850, "50M50N50M")],
    "read_runs_off_contig_end": [rd(5900, "50M100N60M")],
    "refskip_runs_off_contig_end": [rd(5900, "50M200N50M")],
    "cigar_starts_with_refskip": [dict(pos=1000, cigar="100N50M", seq=G[1100:1150], xs="+")],
    "cigar_ends_with_refskip": [dict(pos=1000, cigar="50M100N", seq=G[1000:1050], xs="+")],
    "softclip_longer_than_read": [dict(pos=1000, cigar="60S50M100N50M", seq="ACGT" * 10, xs="+")],
    "sequence_shorter_than_cigar": [dict(pos=1000, cigar="50M100N50M", seq="ACGT" * 10, xs="+")],
    "bad_xs_value": [dict(pos=1000, cigar="50M100N50M", seq=G[1000:1050] + G[1150:1200], xs="*")],
    "bad_xs_on_unspliced_read": [dict(pos=900, cigar="50M", seq=None, xs="x"), rd(1000, "50M100N50M")],
    "secondary_supplementary_qcfail_dup": [rd(1000, "50M100N50M", flag=0x100 | 0x800 | 0x200 | 0x400), rd(1000, "50M100N50M", flag=4)],
    "mapq_threshold": [rd(1000, "50M100N50M", mapq=30), rd(1001, "49M100N51M", mapq=29), rd(1002, "48M100N52M", mapq=255)],
    "xs_mixture_below_threshold": [rd(1000 + i, f"{50 - i}M100N{50 + i}M", xs=("+" if i < 18 else "-")) for i in range(19)]
    + [rd(1025, "25M100N75M", xs=None)],
    "xs_exactly_95_percent": [rd(1000 + i, f"{50 - i}M100N{50 + i}M", xs=("+" if i < 19 else None)) for i in range(20)],
    "many_reads_same_position_interleaved_ends": [rd(1000, c) for c in ["50M100N50M", "50M100N40M", "50M100N50M", "50M100N40M", "50M100N30M"]],
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_case(ffi, orc, name):
    reads = sorted(CASES[name], key=lambda r: r["pos"])
    both(ffi, orc, G, reads)
    both(ffi, orc, G, reads, orientation="FR")


def test_exotic_genome_characters(ffi, orc):
    """'X' in the contig (query padding really matches it), characters outside the nt16 alphabet,
    soft-masked and IUPAC bases: the byte-wise path of k4 must give the oracle's numbers."""
    g = list(G)
    for p, c in ((995, "X"), (1011, "x"), (1105, "X"), (1120, "*"), (1160, "-"), (1170, "n"), (1171, "R"), (1172, "y"), (1030, "=")):
        g[p] = c
    for p in range(1180, 1200):
        g[p] = g[p].lower()
    genome = "".join(g)
    # the third junction window [985, 1049] encloses the small intron [990, 1009] of the first read, whose
    # 'X' padding meets a real 'X' in the contig at 995
    reads = [read_from_genome(G, 975, "15M20N40M100N30M"), read_from_genome(G, 985, "65M100N15M"),
             read_from_genome(G, 1000, "50M100N50M"), read_from_genome(G, 1001, "20M3D26M100N51M")]
    status, rows = both(ffi, orc, genome, reads)
    assert status == "ok" and rows["sum_mismatches"].sum() > 0


def test_long_read_many_ops(ffi, orc):
    """CIGAR with more ops than the LDS staging holds (long reads): ops beyond the 8th come from global memory."""
    parts, pos = [], 500
    cig = ""
    for k in range(12):
        cig += f"{20 + k}M{3 + (k % 3)}{'ID'[k % 2]}{15}M{60 + 7 * k}N"
    cig += "40M"
    reads = [read_from_genome(G, 500, cig), read_from_genome(G, 520, "16M60N40M")]
    status, rows = both(ffi, orc, G, reads)
    assert status == "ok" and len(rows) >= 12


def test_chain_on_a_context_made_without_chain_slots(ffi, orc):
    """PJB_FLAG_NO_CHAINS only postpones the chain slots' streams: a chain on such a context gives the same rows."""
    reads = [read_from_genome(G, 300, "50M100N50M"), read_from_genome(G, 320, "30M100N70M"), read_from_genome(G, 900, "40M200N60M")]
    b = ReadBatch.from_reads(reads)
    orows, oreg = orc.find_juncs(0, len(G), G, b, "UNKNOWN")
    with ffi.Context(0, "UNKNOWN", flags=ffi.FLAG_NO_CHAINS) as ctx:
        ctx.set_refs([len(G)])
        for _ in range(2):
            drows, dreg = ffi.run_contig(ctx, 0, G.encode(), [b])
            region_equal(dreg, oreg)
            assert_rows_equal(drows, orows)
            ctx.clear_rows()


@pytest.mark.parametrize("seed", [7002, 7003, 7006, 7009])
def test_fuzz_family_many_ops(ffi, orc, seed):
    """Round 4's verdict: a fuzz family of long CIGARs (reads of hundreds of bases over several introns, an indel / = / X / P run every
    few bases, clips): tens of operations a read, up to ~70 -- far past the eight operations the kernels keep close at hand."""
    import re

    from fuzzgen import make_reads
    genome, reads = make_reads(seed, glen=200000, n_reads=300, paired=True, opts=dict(indel=0.92, eqx=0.05, pad=0.02, sub=0.01, clip=0.3, hard=0.05),
                               L=(1500, 4000), n_tx=8, noseq_frac=0.0)
    n_ops = sorted(len(re.findall(r"\d+[MIDNSHP=X]", r["cigar"])) for r in reads)
    assert n_ops[len(n_ops) // 2] >= 25 and n_ops[-1] >= 50, (n_ops[len(n_ops) // 2], n_ops[-1])
    status, rows = both(ffi, orc, genome, reads, "FR")
    assert status == "ok" and len(rows) > 0


@pytest.mark.parametrize("seed,ori", [(9001, "FR"), (9002, "UNKNOWN"), (9003, "RF"), (9004, "FF"), (9005, "SE"), (9006, "FR"), (9007, "UNKNOWN"), (9008, "FR")])
def test_fuzz_family_exception_channel(ffi, orc, seed, ori):
    """Round 5's verdict: the inputs of the 2-bit compare's exception channel.  A genome that is pure ACGT but for letters planted at
    the anchors' edges and at the boundaries of the exception bitmap's 64-base stretches (N, IUPAC codes, lower case), reads with N /
    IUPAC codes / '=' of their own at their anchors' first and last bases: the lanes of one wavefront take the 2-bit rounds, fall back
    to the 4-bit codes under a flagged stretch, or never leave them -- and every row must equal the oracle's.  Then the same records
    with 4-bit bases only (pjb_batch.seq2 = NULL): the same rows."""
    from fuzzgen import EXC_OPTS, make_reads
    from portcullis_amd.records import pack_seq2
    genome, reads = make_reads(seed, glen=30000 + 997 * (seed % 7), n_reads=2500, paired=ori != "SE", opts=EXC_OPTS, L=(40, 260))
    b = ReadBatch.from_reads(reads)
    _, sx = pack_seq2(b.seq4, b.seq_off, b.l_qseq)
    spliced = np.array(["N" in r["cigar"] for r in reads])
    marked = np.unpackbits(sx.view(np.uint8), bitorder="little")[: b.n].astype(bool) & spliced
    assert 0.05 < marked.sum() / spliced.sum() < 0.8 and sum(c.upper() not in "ACGT" for c in genome) >= 5  # both kinds of lanes side by side
    status, orows = both(ffi, orc, genome, reads, ori)
    assert status == "ok" and len(orows) > 10
    with ffi.Context(0, ori) as ctx:
        ctx.set_refs([len(genome)])
        ctx.upload_contig(0, genome.encode())
        ctx.submit_batch(0, b, seq2=False)
        ctx.finish_contig(0)
        assert_rows_equal(ctx.collect(), orows)


def test_exception_bitmap_at_stretch_and_word_boundaries(ffi, orc):
    """One letter outside ACGT moved base by base across a read's anchors, over the boundaries of the genome's 64-base stretches and
    of the bitmap's 32-stretch words: under an anchor it is a mismatch (the lane falls back to the 4-bit codes), next to it nothing."""
    rng = np.random.default_rng(77)
    g0 = "".join(rng.choice(list("ACGT"), size=9000))
    for pos in (2048 - 40, 4096 - 75, 64 * 31 - 10, 2048 * 2 + 3):
        for at in list(range(pos - 2, pos + 3)) + list(range(pos + 38, pos + 43)) + list(range(pos + 138, pos + 143)) + [pos + 140 + 59, pos + 140 + 60]:
            for ch in ("NRnc"[at % 4],):  # ('c': lower case only -- upper-cased it is no exception)
                g = g0[:at] + ch + g0[at + 1:]
                reads = [read_from_genome(g0, pos - 3, "5S43M100N20M50N37M4S"), read_from_genome(g0, pos, "40M100N60M"), read_from_genome(g0, pos + 1, "39M100N61M")]
                status, _ = both(ffi, orc, g, reads)
                assert status == "ok"
