"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle
on the same seeded inputs, plus the committed reference fixtures."""
import os

import numpy as np
import pytest

from fixtures_micro import micro1, micro2
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


def run_both(ffi, orc, genome, batch, orientation="UNKNOWN", split=None, ref_len=None):
    ref_len = ref_len or len(genome)
    orows, oreg = orc.find_juncs(0, ref_len, genome, batch, orientation)
    with ffi.Context(0, orientation) as ctx:
        ctx.set_refs([ref_len])
        if split is not None:
            cuts = [0] + sorted(set(int(x) for x in split if 0 < x < batch.n)) + [batch.n]
            batches = [batch.slice(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
        else:
            batches = [batch]
        drows, dreg = ffi.run_contig(ctx, 0, genome.encode() if isinstance(genome, str) else genome, batches)
    region_equal(dreg, oreg)
    assert dreg["n_junctions"] == len(orows)
    return assert_rows_equal(drows, orows), drows, orows


def test_micro_fixture_1(ffi, orc, spombe30k):
    _, genome = spombe30k
    _, drows, _ = run_both(ffi, orc, genome, ReadBatch.from_reads(micro1(genome)))
    # spot-check the reference-recorded values directly on the device rows
    d = {(int(r["start"]), int(r["end"])): r for r in drows}
    j2 = d[(1170, 1369)]
    assert j2["nb_raw"] == 3 and j2["sum_mismatches"] == 103 and f"{j2['entropy']:g}" == "0.918296"
    assert list(j2["jad"][:10]) == [3] * 10 and list(j2["jad"][10:]) == [2] * 10


def test_micro_fixture_2(ffi, orc, spombe30k):
    _, genome = spombe30k
    for ori in ("FR", "UNKNOWN", "RF", "FF", "SE"):
        run_both(ffi, orc, genome, ReadBatch.from_reads(micro2(genome)), ori)


def test_clipped3(ffi, orc, golden_dir):
    from util_bam import read_bam, records_to_batch
    refs, recs = read_bam(os.path.join(golden_dir, "clipped3.bam"))
    batch = records_to_batch([r for r in recs if r["tid"] == 0])
    rng = np.random.default_rng(4)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=refs[0][1]).tobytes()
    _, drows, _ = run_both(ffi, orc, genome, batch, ref_len=refs[0][1])
    r = drows[0]
    # reference row, SURVEY App. B (genome independent columns)
    assert (r["start"], r["end"], r["left"], r["right"]) == (6442658, 6442841, 6442559, 6442940)
    assert (r["nb_raw"], r["nb_dist"], r["nb_um"], r["nb_bpp"], r["nb_rel"]) == (135, 37, 131, 83, 131)
    assert f"{r['entropy']:g}" == "4.89824"


@pytest.mark.parametrize("seed", range(12))
def test_fuzz(ffi, orc, seed):
    genome, reads = make_reads(seed, n_reads=2500, paired=(seed % 2 == 1))
    ori = ["UNKNOWN", "FR", "RF", "FF"][seed % 4]
    run_both(ffi, orc, genome, to_batch(reads), ori)


@pytest.mark.parametrize("seed", [100, 101, 102])
def test_fuzz_multibatch(ffi, orc, seed):
    genome, reads = make_reads(seed, n_reads=4000, paired=True)
    b = to_batch(reads)
    rng = np.random.default_rng(seed)
    run_both(ffi, orc, genome, b, "FR", split=rng.integers(1, b.n, size=5))


def test_deep_junction(ffi, orc):
    """One junction with tens of thousands of alignments (fragment reduction across many wavefronts)."""
    genome, reads = make_reads(7, n_reads=60000, n_tx=2, L=(60, 110))
    run_both(ffi, orc, genome, to_batch(reads))


def test_no_spliced_reads(ffi, orc):
    rng = np.random.default_rng(1)
    genome = "".join(rng.choice(list("ACGT"), size=5000))
    reads = [dict(pos=int(p), cigar="50M", seq=None, flag=0) for p in sorted(rng.integers(0, 4000, size=300))]
    ent, drows, orows = run_both(ffi, orc, genome, ReadBatch.from_reads(reads))
    assert len(drows) == 0


def test_empty_contig(ffi):
    with ffi.Context(0) as ctx:
        ctx.set_refs([1000])
        reg = ctx.finish_contig(0)
        assert reg["n_reads"] == 0 and reg["min_len"] == 2**31 - 1 and reg["max_len"] == 0
        assert len(ctx.collect()) == 0


def test_error_unsorted(ffi):
    reads = [dict(pos=500, cigar="20M100N20M", seq="A" * 40), dict(pos=100, cigar="20M100N20M", seq="A" * 40)]
    with ffi.Context(0) as ctx:
        ctx.set_refs([5000])
        ctx.upload_contig(0, b"A" * 5000)
        ctx.submit_batch(0, ReadBatch.from_reads(reads))
        with pytest.raises(ffi.PjbError) as e:
            ctx.finish_contig(0)
        assert e.value.code == -14


def test_error_cigar_ends_with_refskip(ffi, orc):
    """CIGAR ending in N: the reference throws in getPaddedQuerySeq (bam_alignment.cc:342)."""
    genome = "ACGT" * 500
    reads = [dict(pos=100, cigar="30M100N", seq="A" * 30, xs="+")]
    b = ReadBatch.from_reads(reads)
    with pytest.raises(orc.OracleError):
        orc.find_juncs(0, len(genome), genome, b, "UNKNOWN")
    with ffi.Context(0) as ctx:
        ctx.set_refs([len(genome)])
        ctx.upload_contig(0, genome.encode())
        ctx.submit_batch(0, b)
        with pytest.raises(ffi.PjbError):
            ctx.finish_contig(0)


def test_seq_star_fallback(ffi, orc):
    """SEQ '*' (l_qseq 0): calcMatchStats fallback branch, junction.cc:168-185."""
    genome = ("ACGTTGCA" * 400)
    reads = [dict(pos=100, cigar="30M100N40M", seq=None, xs="+"), dict(pos=110, cigar="20M100N45M", seq=None, xs="-")]
    run_both(ffi, orc, genome, ReadBatch.from_reads(reads))


def _donor_with_acceptors(n_acc, depth=6):
    rng = np.random.default_rng(n_acc)
    genome = "".join(rng.choice(list("ACGT"), size=20000))
    reads = []
    for k in range(depth):
        for a in range(n_acc):  # one donor at 1040, acceptors 100 + 37 a bases on; plus an ordinary junction further along
            reads.append(dict(pos=1000 + k, cigar=f"{40 - k}M{100 + 37 * a}N{30 + k}M", seq="A" * 70, xs="+", flag=0))
    for k in range(20):
        reads.append(dict(pos=9000 + k, cigar=f"{50 - k}M500N{20 + k}M", seq="C" * 70, xs="-", flag=0))
    reads.sort(key=lambda r: r["pos"])
    return genome, reads


@pytest.mark.parametrize("n_acc", [8, 9, 40])
def test_dense_ids_many_acceptors(ffi, orc, n_acc):
    """K2d keeps 8 intron ends per intron start; a donor with more alternative acceptors sends the contig through
    the full-key sort instead (same rows either way)."""
    genome, reads = _donor_with_acceptors(n_acc)
    batch = ReadBatch.from_reads(reads)
    orows, oreg = orc.find_juncs(0, len(genome), genome, batch, "UNKNOWN")
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([len(genome)])
        drows, dreg = ffi.run_contig(ctx, 0, genome.encode(), [batch])
        passes = ctx.timing()["sort_passes"]
    region_equal(dreg, oreg)
    assert_rows_equal(drows, orows)
    assert len(drows) == n_acc + 1
    assert (passes <= 2) == (n_acc <= 8), passes  # ids: 12 bits; keys: 15 + 18 bits


def test_dense_ids_off(ffi, orc):
    """pjb_set_option("dense_ids", 0): the round-1 sort of the full intron keys."""
    genome, reads = make_reads(5, n_reads=2500, paired=True)
    batch = to_batch(reads)
    orows, oreg = orc.find_juncs(0, len(genome), genome, batch, "FR")
    with ffi.Context(0, "FR") as ctx:
        ctx.set_option("dense_ids", 0)
        ctx.set_refs([len(genome)])
        drows, dreg = ffi.run_contig(ctx, 0, genome.encode(), [batch])
        assert ctx.timing()["sort_passes"] >= 3
    region_equal(dreg, oreg)
    assert_rows_equal(drows, orows)


def test_dense_ids_junction_limit(ffi, orc):
    """More junctions than the context planned for (limit: pairs / 32, at least 4096): the control block reports the
    overflow and the contig is queued again with the count it found."""
    rng = np.random.default_rng(11)
    glen = 400000
    genome = "".join(rng.choice(list("ACGT"), size=glen))
    reads = [dict(pos=20 + 60 * k, cigar=f"25M{30 + (k % 7)}N25M", seq="G" * 50, xs="+", flag=0) for k in range(6000)]
    batch = ReadBatch.from_reads(reads)
    orows, oreg = orc.find_juncs(0, glen, genome, batch, "UNKNOWN")
    assert len(orows) == 6000
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([glen])
        drows, dreg = ffi.run_contig(ctx, 0, genome.encode(), [batch])
        region_equal(dreg, oreg)
        assert_rows_equal(drows, orows)
        # the context remembers: the same contig again settles at once
        ctx.clear_rows()
        drows2, _ = ffi.run_contig(ctx, 0, genome.encode(), [batch])
        assert drows2.tobytes() == drows.tobytes()


def _three_contigs(orc, seeds=(21, 22, 23)):
    out = []
    for tid, seed in enumerate(seeds):
        genome, reads = make_reads(seed, n_reads=2500, paired=True)
        batch = to_batch(reads)
        orows, oreg = orc.find_juncs(tid, len(genome), genome, batch, "FR")
        out.append((genome, batch, orows, oreg))
    return out


def _run_queued(ffi, ctx, contigs, depth):
    """Contigs through pjb_finish_contig_begin / _end with `depth` contigs queued (1: plain finish order)."""
    ctx.set_refs([len(g) for g, _, _, _ in contigs])
    for tid, (g, _, _, _) in enumerate(contigs):
        ctx.upload_contig(tid, g.encode() if isinstance(g, str) else g)
    ctx.clear_rows()
    regs, queued = {}, []
    for tid, (_, b, _, _) in enumerate(contigs):
        ctx.submit_batch(tid, b)
        ctx.finish_contig_begin(tid)
        queued.append(tid)
        if len(queued) >= depth:
            t = queued.pop(0)
            regs[t] = ctx.finish_contig_end(t)
    for t in queued:
        regs[t] = ctx.finish_contig_end(t)
    return ctx.collect(), regs


def test_two_contigs_queued(ffi, orc):
    """pjb_finish_contig_begin / _end: the second contig is queued before the first is collected; its rows follow the
    first one's through the device-side row cursor.  Same table as finishing one contig at a time."""
    contigs = _three_contigs(orc)
    want = np.concatenate([c[2] for c in contigs])
    with ffi.Context(0, "FR") as ctx:
        for depth in (2, 1, 3, ffi.MAX_QUEUED):
            rows, regs = _run_queued(ffi, ctx, contigs, depth)
            for tid, c in enumerate(contigs):
                region_equal(regs[tid], c[3])
            assert_rows_equal(rows, want)
        # out of order / too many
        ctx.clear_rows()
        ctx.submit_batch(0, contigs[0][1])
        ctx.submit_batch(1, contigs[1][1])
        ctx.submit_batch(2, contigs[2][1])
        ctx.finish_contig_begin(0)
        ctx.finish_contig_begin(1)
        with pytest.raises(ffi.PjbError):
            ctx.finish_contig_begin(1)
        with pytest.raises(ffi.PjbError):
            ctx.finish_contig_end(1)
        with pytest.raises(ffi.PjbError):
            ctx.clear_rows()
        ctx.finish_contig_end(0)
        ctx.finish_contig_begin(2)
        ctx.finish_contig_end(1)
        ctx.finish_contig_end(2)
        assert_rows_equal(ctx.collect(), want)
    with ffi.Context(0, "FR") as ctx:  # more than the queue holds (targets without alignments count as well)
        ctx.set_refs([1000] * (ffi.MAX_QUEUED + 1))
        for tid in range(ffi.MAX_QUEUED):
            ctx.finish_contig_begin(tid)
        with pytest.raises(ffi.PjbError):
            ctx.finish_contig_begin(ffi.MAX_QUEUED)
        for tid in range(ffi.MAX_QUEUED):
            assert ctx.finish_contig_end(tid)["n_reads"] == 0
        ctx.finish_contig_begin(ffi.MAX_QUEUED)
        ctx.finish_contig_end(ffi.MAX_QUEUED)


def test_queued_behind_an_overflow(ffi, orc):
    """The first of two queued contigs exceeds the junction limit it was queued with: it is repeated, and the contig
    queued behind it (whose rows went to the wrong place meanwhile) is queued again."""
    rng = np.random.default_rng(11)
    glen = 400000
    genome = "".join(rng.choice(list("ACGT"), size=glen))
    reads = [dict(pos=20 + 60 * k, cigar=f"25M{30 + (k % 7)}N25M", seq="G" * 50, xs="+", flag=0) for k in range(6000)]
    big = ReadBatch.from_reads(reads)
    orows_big, oreg_big = orc.find_juncs(0, glen, genome, big, "UNKNOWN")
    g2, reads2 = make_reads(31, n_reads=2500)
    b2 = to_batch(reads2)
    orows2, oreg2 = orc.find_juncs(1, len(g2), g2, b2, "UNKNOWN")
    contigs = [(genome, big, orows_big, oreg_big), (g2, b2, orows2, oreg2)]
    with ffi.Context(0, "UNKNOWN") as ctx:
        rows, regs = _run_queued(ffi, ctx, contigs, 2)
        region_equal(regs[0], oreg_big)
        region_equal(regs[1], oreg2)
        assert_rows_equal(rows, np.concatenate([orows_big, orows2]))


def test_queued_behind_an_error(ffi, orc):
    """The first of two queued contigs fails (unsorted records): its end reports the error, the second contig still
    yields its rows."""
    g1 = "ACGT" * 2000
    bad = ReadBatch.from_reads([dict(pos=500, cigar="30M100N40M", seq="A" * 70, xs="+"), dict(pos=100, cigar="70M", seq=None)])
    g2, reads2 = make_reads(32, n_reads=2500)
    b2 = to_batch(reads2)
    orows2, oreg2 = orc.find_juncs(1, len(g2), g2, b2, "UNKNOWN")
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([len(g1), len(g2)])
        ctx.upload_contig(0, g1.encode())
        ctx.upload_contig(1, g2.encode())
        ctx.clear_rows()
        ctx.submit_batch(0, bad)
        ctx.submit_batch(1, b2)
        ctx.finish_contig_begin(0)
        ctx.finish_contig_begin(1)
        with pytest.raises(ffi.PjbError) as e:
            ctx.finish_contig_end(0)
        assert e.value.code == -14
        reg = ctx.finish_contig_end(1)
        region_equal(reg, oreg2)
        assert_rows_equal(ctx.collect(), orows2)


def test_long_reads(ffi, orc):
    """Reads longer than the 160 bases k4a_simple stages per read in LDS (per-lane loads instead), mixed with short ones."""
    genome, reads = make_reads(41, n_reads=3000, paired=True, L=(120, 400))
    run_both(ffi, orc, genome, to_batch(reads), "FR")


def test_options_do_not_change_rows(ffi, orc):
    """pjb_set_option: kernels of a chain on one stream instead of several ("overlap" 0), the sort on the full keys
    instead of the dense ids ("dense_ids" 0) -- the rows stay what they are; options need an empty queue."""
    contigs = _three_contigs(orc, seeds=(51, 52, 53))
    want = np.concatenate([c[2] for c in contigs])
    with ffi.Context(0, "FR") as ctx:
        for overlap, dense in ((0, 1), (1, 0), (0, 0), (1, 1)):
            ctx.set_option("overlap", overlap)
            ctx.set_option("dense_ids", dense)
            rows, _ = _run_queued(ffi, ctx, contigs, 3)
            assert_rows_equal(rows, want)
        with pytest.raises(ffi.PjbError):
            ctx.set_option("no_such_option", 1)
        ctx.clear_rows()
        ctx.submit_batch(0, contigs[0][1])
        ctx.finish_contig_begin(0)
        with pytest.raises(ffi.PjbError):
            ctx.set_option("overlap", 0)
        ctx.finish_contig_end(0)


def test_row_mirror(ffi, orc):
    """pjb_set_row_mirror: finish_contig leaves { n_rows, spliced, unspliced, sum_len, min_len, max_len } and the rows
    in the caller's device buffer -- also for a contig without junctions -- and refuses a buffer that is too small."""
    import torch
    torch.zeros(1, device="cuda")  # torch brings its own HIP runtime: it must come up before the library's
    genome, reads = make_reads(5, n_reads=2000)
    batch = to_batch(reads)
    unspliced = to_batch([r for r in reads if "N" not in r["cigar"]][:50]) if isinstance(reads[0]["cigar"], str) else None
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([len(genome)])
        buf = torch.zeros(64 + 4000 * ffi.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        ctx.set_row_mirror(buf.data_ptr(), buf.numel())
        rows, reg = ffi.run_contig(ctx, 0, genome.encode(), [batch])
        h = buf.cpu().numpy()
        hdr = h[:48].view(np.int64)
        assert list(hdr) == [len(rows), reg["spliced"], reg["unspliced"], reg["sum_len"], reg["min_len"], reg["max_len"]]
        assert h[64:64 + rows.nbytes].tobytes() == rows.tobytes()
        if unspliced is not None and unspliced.n:
            ctx.clear_rows()
            ctx.submit_batch(0, unspliced)
            reg2 = ctx.finish_contig(0)
            hdr = buf.cpu().numpy()[:48].view(np.int64)
            assert reg2["n_junctions"] == 0 and list(hdr) == [0, 0, reg2["unspliced"], reg2["sum_len"], reg2["min_len"], reg2["max_len"]]
        # the mirror accumulates: two contigs finished without a clear in between -> rows appended, counters folded
        genome_b, reads_b = make_reads(6, n_reads=1500)
        batch_b = to_batch(reads_b)
        ctx.set_refs([len(genome), len(genome_b)])
        ctx.upload_contig(0, genome.encode())
        ctx.upload_contig(1, genome_b.encode())
        ctx.clear_rows()
        ctx.set_row_mirror(buf.data_ptr(), buf.numel())
        ctx.submit_batch(0, batch)
        ra = ctx.finish_contig(0)
        ctx.submit_batch(1, batch_b)
        rb = ctx.finish_contig(1)
        both = ctx.collect()
        h = buf.cpu().numpy()
        assert list(h[:48].view(np.int64)) == [len(both), ra["spliced"] + rb["spliced"], ra["unspliced"] + rb["unspliced"],
                                              ra["sum_len"] + rb["sum_len"], min(ra["min_len"], rb["min_len"]),
                                              max(ra["max_len"], rb["max_len"])]
        assert len(both) == ra["n_junctions"] + rb["n_junctions"] and set(both["refid"]) == {0, 1}
        assert h[64:64 + both.nbytes].tobytes() == both.tobytes()
        # the same with both contigs queued (pjb_finish_contig_begin / _end): the second contig's place in the mirror
        # follows from the device-side cursor
        ctx.clear_rows()
        ctx.set_row_mirror(buf.data_ptr(), buf.numel())
        buf.zero_()
        torch.cuda.synchronize()
        ctx.submit_batch(0, batch)
        ctx.submit_batch(1, batch_b)
        ctx.finish_contig_begin(0)
        ctx.finish_contig_begin(1)
        qa = ctx.finish_contig_end(0)
        qb = ctx.finish_contig_end(1)
        assert qa == ra and qb == rb
        assert ctx.collect().tobytes() == both.tobytes()
        h2 = buf.cpu().numpy()
        assert h2[:48].tobytes() == h[:48].tobytes() and h2[64:64 + both.nbytes].tobytes() == both.tobytes()
        ctx.set_refs([len(genome)])
        ctx.upload_contig(0, genome.encode())
        small = torch.zeros(64 + ffi.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        ctx.set_row_mirror(small.data_ptr(), small.numel())
        ctx.clear_rows()
        ctx.submit_batch(0, batch)
        with pytest.raises(ffi.PjbError):
            ctx.finish_contig(0)
        ctx.set_row_mirror(0, 0)
        rows3, _ = ffi.run_contig(ctx, 0, genome.encode(), [batch])
        assert rows3.tobytes() == rows.tobytes()


def _fasta_lines(genome, width, term):
    return term.join(genome[i:i + width] for i in range(0, len(genome), width))


@pytest.mark.gpu
def test_upload_contig_fasta(ffi, orc):
    """pjb_upload_contig_fasta: the record's bytes as they are in the file (LF and CRLF line ends, a short last line, a
    length that is a multiple of the line width, soft-masked bases) give the rows pjb_upload_contig gives; a record that
    is not laid out as stated is refused without touching the contig."""
    genome, reads = make_reads(61, n_reads=1500)
    batch = to_batch(reads)
    masked = "".join(ch.lower() if (i // 37) % 3 == 0 else ch for i, ch in enumerate(genome))
    orows, oreg = orc.find_juncs(0, len(genome), genome, batch, "UNKNOWN")
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([len(genome)])
        for width, term in ((60, "\n"), (70, "\r\n"), (len(genome), "\n"), (61, "\n")):
            raw = _fasta_lines(masked, width, term).encode()
            assert ctx.upload_contig_fasta(0, raw, width, width + len(term), len(genome))
            ctx.clear_rows()
            ctx.submit_batch(0, batch)
            reg = ctx.finish_contig(0)
            region_equal(reg, oreg)
            assert_rows_equal(ctx.collect(), orows)
        cut = len(genome) - len(genome) % 50  # a whole number of lines
        raw = _fasta_lines(masked[:cut], 50, "\n").encode()
        assert ctx.upload_contig_fasta(0, raw, 50, 51, cut)
        # malformed: a line one base longer than the index says, a blank inside a line, too few bytes
        good = _fasta_lines(masked, 60, "\n")
        ctx.upload_contig(0, genome.encode())
        for bad in (good[:200] + "A" + good[200:], good[:300] + " " + good[301:], good[:-5]):
            assert not ctx.upload_contig_fasta(0, bad.encode(), 60, 61, len(genome))
        ctx.clear_rows()  # the contig uploaded before the refused ones is still the one in use
        ctx.submit_batch(0, batch)
        reg = ctx.finish_contig(0)
        assert_rows_equal(ctx.collect(), orows)


def test_abi3_caller_is_still_served():
    """A caller compiled against ABI 3 (pjb_batch ends at name_hash, pjb_timing at checked_reads): the library must not read the two pointers
    ABI 4 added to the batch -- the stub puts a wild address there -- nor write the two counters it added to the timing; the rows are the
    oracle's (the compares run on the 4-bit bases)."""
    from fuzzgen import make_reads, to_batch
    from oracle import oracle as orc
    from parity import assert_rows_equal, region_equal
    from portcullis_amd import ffi

    genome, reads = make_reads(4242, n_reads=2500, paired=True)
    batch = to_batch(reads)
    orows, oreg = orc.find_juncs(0, len(genome), genome, batch, "FR")
    with ffi.Context(0, "FR", abi_version=3) as ctx:
        ctx.set_refs([len(genome)])
        drows, dreg = ffi.run_contig(ctx, 0, genome.encode(), [batch])
        region_equal(dreg, oreg)
        assert_rows_equal(drows, orows)
        t = ctx.timing()
        assert t["repeats"] == -77 and t["repeat_reasons"] == -77 and t["sort_passes"] >= 1
    with pytest.raises(ffi.PjbError):
        ffi.Context(0, "FR", abi_version=2)


def test_seq2_must_be_word_aligned():
    """pjb_batch.seq2 is read as 32-bit words: a device batch whose seq2 starts on an odd 16-bit granule is refused (PJB_ERR_ARG), not misread."""
    import torch
    from portcullis_amd import ffi, synth

    d = synth.generate(synth.CONFIGS["C2-tiny"], device="cuda")
    b = dict(d["batch"])
    odd = torch.zeros(b["seq2"].numel() + 1, dtype=torch.int16, device="cuda")
    odd[1:] = b["seq2"]
    b["seq2"] = odd[1:]  # (2 bytes past a 4-byte boundary)
    assert b["seq2"].data_ptr() % 4 == 2
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([d["config"].contig_len])
        ctx.upload_contig_device(0, d["genome"])
        with pytest.raises(ffi.PjbError) as e:
            ctx.submit_batch_device(0, b, d["n_reads"])
        assert e.value.code == -16
