"""Host BAM reader / writer (no GPU) under AddressSanitizer + UBSan: BamReader::next()/current(), rewind()/nextRecord()
and BamWriter with its in-process .bai, on files whose records straddle BGZF blocks."""
import gzip
import os
import subprocess

import pytest

from fuzzgen import make_reads
from util_bam import write_bam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    host = os.path.join(ROOT, "portcullis_amd", "host")
    out = str(tmp_path_factory.mktemp("asan") / "bam_roundtrip")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           f"-I{host}/include", f"-I{ROOT}/include", "-o", out, os.path.join(ROOT, "tests", "cpp", "bam_roundtrip.cc"),
                           os.path.join(host, "src", "bam_reader.cc"), os.path.join(host, "src", "fast_inflate.cc"), os.path.join(host, "src", "bam_writer.cc"),
                           "-lz", "-lpthread"])
    return out


@pytest.mark.parametrize("block_size,threads", [(0xFF00, 1), (1500, 3)])
def test_roundtrip_under_sanitizers(tmp_path, exe, block_size, threads):
    reads, refs = [], []
    for tid, seed in enumerate([91, None, 92]):
        if seed is None:
            refs.append((f"empty{tid}", 3000))
            continue
        genome, rr = make_reads(seed, n_reads=3000, paired=tid == 2)
        for k, r in enumerate(rr):
            r["tid"] = tid
            r["name"] = f"n{tid}_{k}"
            if r.get("mtid", -1) >= 0:
                r["mtid"] = tid
        refs.append((f"chr{tid + 1}", len(genome)))
        reads += rr
    for k in range(3):
        reads.append(dict(tid=-1, pos=-1, cigar="", seq="ACGT" * 5, flag=4, mapq=0, name=f"u{k}"))
    src = str(tmp_path / "in.bam")
    write_bam(src, refs, reads, block_size=block_size)
    dst = str(tmp_path / "out.bam")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, src, dst, str(threads)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr[-3000:]
    assert f"placed={len(reads) - 3} raw={len(reads)}" in p.stdout
    assert gzip.open(dst, "rb").read() == gzip.open(src, "rb").read()   # same header, same records, same order
    assert os.path.exists(dst + ".bai")


def _records(data):
    import struct
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


@pytest.mark.parametrize("block_size,threads,chunk,drop,ahead", [(0xFF00, 1, 1 << 20, 0, 0), (1500, 3, 1 << 20, 3, 0), (700, 4, 70000, 2, 0),
                                                                 (3000, 2, 1 << 30, 5, 0), (700, 4, 70000, 2, 2), (1500, 3, 200000, 3, 1),
                                                                 (0xFF00, 1, 1 << 20, 0, 3), (700, 4, 70000, 2, "2 async"), (0xFF00, 3, 1 << 20, 0, "0 async"),
                                                                 (3000, 2, 1 << 30, 5, "1 async")])
def test_bulk_route_under_sanitizers(tmp_path, exe, block_size, threads, chunk, drop, ahead):
    """`bamfilt`'s route through the host library: the whole file in pieces (blocks inflated and record starts found by
    several threads, from the index's record starts; unplaced records at the end, which no index covers), the kept records
    gathered and compressed by several threads.  Pieces smaller than the file, records straddling pieces and blocks."""
    reads, refs = [], []
    for tid, seed in enumerate([93, 94, None, 95]):
        if seed is None:
            refs.append((f"empty{tid}", 3000))
            continue
        genome, rr = make_reads(seed, n_reads=2500, paired=tid == 1)
        for k, r in enumerate(rr):
            r["tid"] = tid
            r["name"] = f"n{tid}_{k}"
            if r.get("mtid", -1) >= 0:
                r["mtid"] = tid
        refs.append((f"chr{tid + 1}", len(genome)))
        reads += rr
    for k in range(40):
        reads.append(dict(tid=-1, pos=-1, cigar="", seq="ACGT" * 25, flag=4, mapq=0, name=f"unplaced{k}"))
    src = str(tmp_path / "in.bam")
    write_bam(src, refs, reads, block_size=block_size)
    dst = str(tmp_path / "out.bam")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, src, dst, str(threads), "bulk", str(chunk), str(drop)] + str(ahead).split(), capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr[-3000:]
    assert f"bulk raw={len(reads)}" in p.stdout
    header, recs = _records(gzip.open(src, "rb").read())
    assert len(recs) == len(reads)
    want = header + b"".join(r for i, r in enumerate(recs) if not drop or i % drop)
    assert gzip.open(dst, "rb").read() == want
    # the index written beside it drives a second pass through the same route
    again = str(tmp_path / "again.bam")
    p = subprocess.run([exe, dst, again, str(threads), "bulk", str(chunk), "0"] + str(ahead).split(), capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr[-3000:]
    assert gzip.open(again, "rb").read() == want


@pytest.mark.parametrize("ahead", [0, 2])
@pytest.mark.parametrize("how", ["truncate_mid_block", "truncate_mid_record", "flip_header", "flip_payload", "bad_isize", "index_lies"])
def test_bulk_route_rejects_damaged_files(tmp_path, exe, how, ahead):
    """Damaged input through the bulk route under the sanitizers: an error (exit code 3, a message), never a crash, a hang
    or a sanitizer report."""
    import shutil
    genome, rr = make_reads(96, n_reads=2000)
    for k, r in enumerate(rr):
        r["tid"] = 0
        r["name"] = f"d{k}"
    src = str(tmp_path / "in.bam")
    write_bam(src, [("chr1", len(genome))], rr, block_size=1200)
    raw = bytearray(open(src, "rb").read())
    bad = str(tmp_path / "bad.bam")
    shutil.copy(src + ".bai", bad + ".bai")
    n = len(raw)
    if how == "truncate_mid_block":
        raw = raw[:n // 2 + 7]
    elif how == "truncate_mid_record":
        # cut at a block boundary in the middle of the file: the last record of what is left is incomplete
        o, cut = 0, 0
        while o < n // 2:
            cut = o
            o += (raw[o + 16] | raw[o + 17] << 8) + 1
        raw = raw[:cut]
    elif how == "flip_header":
        o = 0
        for _ in range(5):
            o += (raw[o + 16] | raw[o + 17] << 8) + 1
        raw[o + 1] ^= 0xFF
    elif how == "flip_payload":
        o = 0
        for _ in range(7):
            o += (raw[o + 16] | raw[o + 17] << 8) + 1
        raw[o + 30] ^= 0x55
        raw[o + 31] ^= 0xAA
    elif how == "bad_isize":
        o = 0
        for _ in range(3):
            o += (raw[o + 16] | raw[o + 17] << 8) + 1
        size = (raw[o + 16] | raw[o + 17] << 8) + 1
        raw[o + size - 4] ^= 0x10
    elif how == "index_lies":
        bai = bytearray(open(bad + ".bai", "rb").read())
        for k in range(len(bai) - 16, 40, -8):   # nudge a few virtual offsets off their record starts
            if bai[k] and k % 24 == 0:
                bai[k] = (bai[k] + 3) & 0xFF
        open(bad + ".bai", "wb").write(bai)
    open(bad, "wb").write(raw)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, bad, str(tmp_path / "out.bam"), "3", "bulk", str(1 << 16), "0", str(ahead)], capture_output=True, text=True, timeout=300, env=env)
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    assert p.returncode in (0, 3), (p.returncode, p.stderr[-2000:])
    if how in ("truncate_mid_block", "flip_header", "bad_isize"):
        assert p.returncode == 3 and "Error:" in p.stderr


@pytest.mark.parametrize("n_reads", [0, 1])
@pytest.mark.parametrize("route", ["0", "2 async"])
def test_bulk_route_on_nearly_empty_files(tmp_path, exe, n_reads, route):
    """A file with a header and no record (or one): nothing for the scan to deliver, nothing for the writer to hand over --
    the output is the header (and the record), the EOF block and an index."""
    refs = [("chr1", 5000), ("chr2", 3000)]
    reads = [dict(tid=0, pos=100, cigar="50M", seq="A" * 50, flag=0, mapq=60, name="r1")][:n_reads]
    src, dst = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    write_bam(src, refs, reads, block_size=1500)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, src, dst, "3", "bulk", "70000", "0"] + route.split(), capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and f"bulk raw={n_reads}" in p.stdout, p.stdout + p.stderr[-3000:]
    assert gzip.open(dst, "rb").read() == gzip.open(src, "rb").read()
    assert os.path.exists(dst + ".bai")
