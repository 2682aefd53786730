"""Host side of the device ingest (no GPU needed): the byte ranges BamReader::regionSpan hands to
pjb_submit_bam start at the block of each target's first record, carry the right inside-block offset,
end with the block in which the target's records end, and readSpan / readRegionBytes return exactly
the file's bytes."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from fuzzgen import make_reads
from util_bam import bai_to_csi, write_bam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    raw = open(path, "rb").read()
    blocks, o = [], 0
    while o < len(raw):
        bs = (raw[o + 16] | raw[o + 17] << 8) + 1
        blocks.append((o, bs, struct.unpack_from("<I", raw, o + bs - 4)[0]))
        o += bs
    ustart = np.cumsum([0] + [b[2] for b in blocks])
    data = gzip.decompress(raw)
    (l_text,) = struct.unpack_from("<i", data, 4)
    p = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, p)
    p += 4
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, p)
        p += 8 + l_name
    first, last = {}, {}
    while p + 4 <= len(data):
        (bs,) = struct.unpack_from("<i", data, p)
        (tid,) = struct.unpack_from("<i", data, p + 4)
        if tid >= 0:
            first.setdefault(tid, p)
            last[tid] = p + 4 + bs  # one past the target's last record
        p += 4 + bs
    return raw, blocks, ustart, first, last, n_ref


@pytest.mark.parametrize("block_size,csi", [(0xFF00, 0), (777, 0), (4096, 1)])
def test_region_spans(tmp_path, block_size, csi):
    host = os.path.join(ROOT, "portcullis_amd", "host")
    csrc = os.path.join(ROOT, "portcullis_amd", "csrc")
    if not os.path.exists(os.path.join(host, "libportcullis_host.so")):
        pytest.skip("host library not built")
    exe = str(tmp_path / "region_span")
    subprocess.check_call(["g++", "-O1", "-std=c++17", f"-I{host}/include", f"-I{ROOT}/include", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "region_span.cc"), f"-L{host}", "-lportcullis_host",
                           f"-L{csrc}", "-lportcullis_amd", f"-Wl,-rpath,{host}", f"-Wl,-rpath,{csrc}"])
    reads, refs = [], []
    for tid, seed in enumerate([5, None, 6, 7]):
        if seed is None:
            refs.append((f"empty{tid}", 2000))
            continue
        genome, rs = make_reads(seed, n_reads=700 + 300 * tid)
        refs.append((f"chr{tid}", len(genome)))
        for k, r in enumerate(rs):
            r["tid"] = tid
            r["name"] = f"t{tid}r{k}"
        reads += rs
    for k in range(900):  # unplaced reads after the last target
        reads.append(dict(tid=-1, pos=-1, cigar="", seq="ACGT" * (1 + k % 30), flag=4, mapq=0, name=f"u{k}"))
    bam = str(tmp_path / "x.bam")
    write_bam(bam, refs, reads, block_size=block_size)
    if csi:
        bai_to_csi(bam + ".bai", bam + ".csi")
        os.remove(bam + ".bai")
    raw, blocks, ustart, first, last, n_ref = parse(bam)
    out = subprocess.run([exe, bam, str(csi)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    assert len(lines) == n_ref
    starts = np.array([b[0] for b in blocks])
    for line in lines:
        f = line.split()
        tid = int(f[0])
        if tid not in first:
            assert f[1] == "none"
            continue
        off, n, first_u, fnv, same = int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5])
        b0 = int(np.searchsorted(ustart, first[tid], side="right") - 1)
        assert off == blocks[b0][0] and first_u == first[tid] - int(ustart[b0])
        assert same == 1
        h = 1469598103934665603
        for x in raw[off:off + n]:
            h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        assert h == fnv
        # the span is made of whole blocks and reaches the end of the target's last record ...
        end = off + n
        assert end == len(raw) or end in set(int(s) for s in starts)
        b_last = int(np.searchsorted(ustart, last[tid] - 1, side="right") - 1)
        assert end >= blocks[b_last][0] + blocks[b_last][1]
        # ... without dragging in the unplaced tail (at most the block after the one in which the target ends)
        assert end <= blocks[min(b_last + 1, len(blocks) - 1)][0] + blocks[min(b_last + 1, len(blocks) - 1)][1]
