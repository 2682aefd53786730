#!/usr/bin/env python3
"""Throughput of the device-side BGZF deflate (run under gpurun): the inflated bytes of the BASELINE configs[1] BAM (2.1 GB of
BAM records) through pjb_deflate_bgzf in pieces of 256 MB; kernel time from HIP events, host-to-host wall time, the size
against the file zlib wrote, zlib -6 on one core on a sample, and the round trip through zlib on a sample.

    python tools/bench_deflate.py [--workdir /tmp/pjb_bamfilt]
"""
import argparse
import json
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workdir", default="/tmp/pjb_bamfilt")
    ap.add_argument("--config", default="C2")
    args = ap.parse_args()
    bam = os.path.join(args.workdir, "prep", "portcullis.sorted.alignments.bam")
    if not os.path.exists(bam):
        e2e = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "e2e_bench.py"), "--config", args.config, "--threads", "16",
                              "--workdir", args.workdir, "--keep", "--no-oracle", "--repeat", "1"], capture_output=True, text=True)
        if e2e.returncode != 0:
            print(e2e.stdout[-2000:], e2e.stderr[-2000:])
            raise SystemExit(1)
    raw = open(bam, "rb").read()
    from portcullis_amd import ffi
    with ffi.Context(0, "UNKNOWN", flags=ffi.FLAG_KERNEL_TIMING) as ctx:
        data = ctx.inflate_bgzf(raw)
        piece = 4096 * 0xff00
        ctx.deflate_bgzf(data[:piece])   # warm-up: allocations
        ctx.reset_kernel_timing()
        t0 = time.perf_counter()
        out_bytes = 0
        first = None
        for o in range(0, len(data), piece):
            blob, sizes = ctx.deflate_bgzf(data[o:o + piece])
            out_bytes += len(blob)
            if first is None:
                first = blob
        wall = time.perf_counter() - t0
        kt = ctx.kernel_timing()
    k = kt["bgzf_deflate"]
    sample = data[:64 * 0xff00]
    t0 = time.perf_counter()
    z = sum(len(zlib.compress(sample[o:o + 0xff00], 6)) for o in range(0, len(sample), 0xff00))
    tz = time.perf_counter() - t0
    import gzip
    assert gzip.decompress(first) == data[:piece]
    print(json.dumps({"input_mb": round(len(data) / 1e6, 1), "blocks": (len(data) + 0xff00 - 1) // 0xff00, "out_mb": round(out_bytes / 1e6, 1),
                      "bam_file_mb": round(len(raw) / 1e6, 1), "ratio_vs_file": round(out_bytes / len(raw), 3),
                      "kernel_ms": round(k[1], 2), "kernel_launches": k[0], "kernel_gbps_input": round(len(data) / (k[1] * 1e-3) / 1e9, 1),
                      "pack_ms": round(kt["bgzf_pack"][1], 2), "host_to_host_wall_s": round(wall, 3),
                      "zlib6_1thread_mbps_input": round(len(sample) / tz / 1e6, 1), "zlib6_sample_ratio_device_over_zlib": None,
                      "round_trip": "first 256 MB piece == gzip.decompress of its members"}))


if __name__ == "__main__":
    main()
