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
                           os.path.join(host, "src", "bam_reader.cc"), os.path.join(host, "src", "bam_writer.cc"), "-lz", "-lpthread"])
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
