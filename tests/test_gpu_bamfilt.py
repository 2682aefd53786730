"""`bamfilt`: the per-alignment keep / drop decision on the device against the oracle's restatement
of BamFilter::filter (src/bam_filter.cc:75-247), and the program end to end: the output BAM holds exactly the records
the oracle keeps, byte for byte as they were in the input."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from fuzzgen import make_reads, to_batch
from util_bam import PREP_BAM, make_prep_dir, read_bam, records_to_batch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "portcullis_amd", "host", "portcullis_amd")


@pytest.fixture(scope="module")
def ffi():
    from portcullis_amd import ffi as f
    assert f.device_count() >= 1
    return f


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    return o


def _junction_subset(orc, genome, batch, rng, keep_frac=0.6):
    rows, _ = orc.find_juncs(0, len(genome), genome, batch, "UNKNOWN")
    keep = rng.random(len(rows)) < keep_frac
    s, e = list(rows["start"][keep]), list(rows["end"][keep])
    # decoys: near misses that no read supports
    for k in range(10):
        s.append(int(rows["start"][k % len(rows)]) + 1)
        e.append(int(rows["end"][k % len(rows)]))
    return np.array(s, dtype=np.int32), np.array(e, dtype=np.int32)


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("mode", ["HARD", "SOFT", "COMPLETE"])
def test_filter_codes_match_oracle(ffi, orc, seed, mode):
    rng = np.random.default_rng(seed)
    genome, reads = make_reads(100 + seed, n_reads=4000, paired=seed % 2 == 0)
    batch = to_batch(reads)
    js_s, js_e = _junction_subset(orc, genome, batch, rng)
    want = orc.bamfilt_flags(batch, js_s, js_e, mode)
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([len(genome)])
        ctx.filter_set_junctions(0, js_s, js_e)
        got = ctx.filter_batch(0, batch, mode)
        assert (got == want).all(), np.nonzero(got != want)[0][:10]
        empty = ctx.filter_batch(1, batch, mode)              # a target without passing junctions: only unspliced survive
        assert (empty == np.where(want == 1, 1, 0)).all()
    assert set(np.unique(want)) >= {0, 1, 2} and ((want == 3).any() or mode == "COMPLETE")
    if mode != "COMPLETE":
        # the reference's walk does not advance over N operations: a read whose SECOND intron passed (and only that one)
        # is dropped, although JunctionSystem::addJunctions had found that junction from this very read
        multi = [i for i, r in enumerate(reads) if r["cigar"].count("N") > 1]
        assert multi and any(want[i] == 0 for i in multi)


def _decompressed(path):
    with gzip.open(path, "rb") as f:
        return f.read()


def _split_records(data):
    (l_text,) = struct.unpack_from("<i", data, 4)
    o = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, o)
    o += 4
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, o)
        o += 8 + l_name
    header, recs = data[:o], []
    while o < len(data):
        (bs,) = struct.unpack_from("<i", data, o)
        recs.append(data[o:o + 4 + bs])
        o += 4 + bs
    return header, recs


@pytest.mark.parametrize("mode,threads,route", [("HARD", 1, "default"), ("COMPLETE", 4, "default"), ("HARD", 3, "zlib"), ("SOFT", 2, "device"),
                                                ("HARD", 4, "handover"), ("SOFT", 1, "handover")])
def test_bamfilt_program(tmp_path, orc, mode, threads, route, monkeypatch):
    if route == "zlib":     # output blocks compressed by zlib on the workers too, the scan in turns with the decisions
        monkeypatch.setenv("PORTCULLIS_HOST_DEFLATE", "1")
        monkeypatch.setenv("PORTCULLIS_SCAN_AHEAD", "0")
    if route == "handover":  # the writer hands over every four blocks: gather beside the flush thread, device deflate per hand-over
        monkeypatch.setenv("PORTCULLIS_FLUSH_BLOCKS", "4")
    if route == "device":   # input blocks inflated on the device, pageable buffers
        monkeypatch.setenv("PORTCULLIS_DEVICE_INFLATE", "1")
        monkeypatch.setenv("PORTCULLIS_PAGEABLE_BUFFERS", "1")
    refs, contigs, reads = [], [], []
    for tid, seed in enumerate([51, 52]):
        genome, rr = make_reads(seed, n_reads=2500, paired=True, glen=20000)
        for k, r in enumerate(rr):
            r["tid"] = tid
            r["name"] = f"q{tid}_{k}"
            if r.get("mtid", -1) >= 0:
                r["mtid"] = tid
        refs.append((f"chr{tid + 1}", len(genome)))
        contigs.append((f"chr{tid + 1}", genome))
        reads += rr
    for k in range(5):   # unplaced reads at the end of the file are kept (not spliced)
        reads.append(dict(tid=-1, pos=-1, cigar="", seq="ACGTACGTAC", flag=4, mapq=0, name=f"u{k}"))
    prep = make_prep_dir(str(tmp_path / "prep"), refs, contigs, reads, block_size=3000)
    bam = os.path.join(prep, PREP_BAM)
    out = str(tmp_path / "junc" / "pc")
    p = subprocess.run([EXE, "junc", "-o", out, prep], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    # "filt": keep two thirds of the junctions
    lines = open(out + ".junctions.tab").read().split("\n")
    body = [l for l in lines[1:] if l.strip()]
    kept = [l for k, l in enumerate(body) if k % 3 != 1]
    tab = str(tmp_path / "pass.junctions.tab")
    open(tab, "w").write("\n".join([lines[0]] + kept) + "\n\n")
    js = {}
    for l in kept:
        c = l.split("\t")
        js.setdefault(int(c[1]), ([], []))
        js[int(c[1])][0].append(int(c[4]))
        js[int(c[1])][1].append(int(c[5]))
    outbam = str(tmp_path / "filt" / "filtered.bam")
    p = subprocess.run([EXE, "bamfilt", "-o", outbam, "-c", mode, "-t", str(threads), "--save_msrs", tab, bam], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    # expected: the input's records, minus the ones the oracle drops
    header, recs_in = _split_records(_decompressed(bam))
    _, parsed = read_bam(bam)
    assert len(parsed) == len(recs_in)
    codes = np.ones(len(parsed), dtype=np.uint8)
    for tid in range(len(refs)):
        idx = [i for i, r in enumerate(parsed) if r["tid"] == tid]
        b = records_to_batch([parsed[i] for i in idx])
        s, e = js.get(tid, ([], []))
        codes[idx] = orc.bamfilt_flags(b, s, e, mode)
    want = header + b"".join(r for r, c in zip(recs_in, codes) if c)
    got = _decompressed(outbam)
    assert got == want
    n_in, n_out, n_mod = len(recs_in), int((codes > 0).sum()), int((codes == 3).sum())
    assert f"Filtered out {n_in - n_out} alignments.  In: {n_in}; Out: {n_out} (Modified: {n_mod});" in p.stdout
    assert n_out < n_in and (n_mod > 0 or mode == "COMPLETE")
    # --save_msrs: the "modified" reads, unmodified in both files (the reference never rewrites the record)
    msr = b"".join(r for r, c in zip(recs_in, codes) if c == 3)
    assert _decompressed(outbam + ".mod.bam") == header + msr and _decompressed(outbam + ".unmod.bam") == header + msr
    # the index the writer made is usable: the filtered file goes through the program again (nothing more to drop)
    again = str(tmp_path / "filt" / "again.bam")
    p2 = subprocess.run([EXE, "bamfilt", "-o", again, "-c", mode, tab, outbam], capture_output=True, text=True, timeout=600)
    assert p2.returncode == 0, p2.stderr[-2000:]
    assert f"In: {n_out}; Out: {n_out}" in p2.stdout
    assert _decompressed(again) == got
    # and its .bai drives a region read: junc on a prep directory whose BAM is the filtered file
    prep2 = str(tmp_path / "prep2")
    os.makedirs(prep2)
    for f in os.listdir(prep):
        if f.startswith("portcullis.genome"):
            os.symlink(os.path.join(prep, f), os.path.join(prep2, f))
    os.symlink(outbam, os.path.join(prep2, PREP_BAM))
    os.symlink(outbam + ".bai", os.path.join(prep2, PREP_BAM + ".bai"))
    out2 = str(tmp_path / "junc2" / "pc")
    p3 = subprocess.run([EXE, "junc", "-o", out2, prep2], capture_output=True, text=True, timeout=600)
    assert p3.returncode == 0, p3.stderr[-2000:]
    n_junc2 = len([l for l in open(out2 + ".junctions.tab").read().split("\n")[1:] if l.strip()])
    assert 0 < n_junc2 <= len(body)
