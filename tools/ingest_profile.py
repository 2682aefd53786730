#!/usr/bin/env python3
"""One target of a prepared BAM through pjb_submit_bam + pjb_finish_contig, a few times (the program rocprofv3 wraps
for the ingest kernels' profile).  usage: ingest_profile.py <prep_dir> [repeats]"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    prep, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from portcullis_amd import ffi
    raw = np.fromfile(os.path.join(prep, "portcullis.sorted.alignments.bam"), dtype=np.uint8)
    # header block(s): find the first record of target 0 by inflating the first block on the host
    import zlib
    b0 = (int(raw[16]) | int(raw[17]) << 8) + 1
    hdr = zlib.decompress(raw[18:b0 - 8].tobytes(), -15)
    (l_text,) = struct.unpack_from("<i", hdr, 4)
    p = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", hdr, p)
    p += 4
    lens = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", hdr, p)
        (l_ref,) = struct.unpack_from("<i", hdr, p + 4 + l_name)
        lens.append(l_ref)
        p += 8 + l_name
    assert p < len(hdr), "header spans several blocks: not handled by this helper"
    genome = open(os.path.join(prep, "portcullis.genome.fa"), "rb").read().split(b"\n")
    seq = b"".join(genome[1:genome.index(next(l for l in genome[1:] if l.startswith(b">")))] if any(l.startswith(b">") for l in genome[1:]) else genome[1:])
    with ffi.Context(0, "UNKNOWN", flags=ffi.FLAG_KERNEL_TIMING) as ctx:
        ctx.set_refs(lens)
        ctx.upload_contig(0, seq.upper())
        for _ in range(reps):
            ctx.clear_rows()
            n = ctx.submit_bam(0, raw, p)
            reg = ctx.finish_contig(0)
        kt = ctx.kernel_timing()
        print(n, "records,", reg["n_junctions"], "junctions")
        for k, (c, ms) in sorted(kt.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"  {k:20s} {c:4d} launches {ms / max(c, 1):8.3f} ms avg")


if __name__ == "__main__":
    main()
