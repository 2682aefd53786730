#!/usr/bin/env python3
"""Throughput of the device-side BGZF inflate on a synthetic prepared BAM (run under gpurun):
builds the C2 BAM with tests/e2e_bench.py, inflates it through pjb_inflate_bgzf in chunks of whole
blocks and reports the kernel rate (HIP events) beside single-thread zlib on the same bytes.

    python tools/bench_inflate.py [--config C2] [--chunk-mb 256]
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


def block_offsets(raw):
    offs, o, n = [], 0, len(raw)
    while o < n:
        offs.append(o)
        o += (raw[o + 16] | raw[o + 17] << 8) + 1  # BSIZE of blocks written with the 6-byte BC extra field only
    offs.append(n)
    return offs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--chunk-mb", type=int, default=256)
    ap.add_argument("--workdir", default="/tmp/pjb_inflate")
    ap.add_argument("--times", type=int, default=1, help="inflate the file's blocks this many times in one call (a larger input: BGZF blocks are independent)")
    args = ap.parse_args()
    e2e = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "e2e_bench.py"), "--config", args.config, "--threads", "16",
                          "--workdir", args.workdir, "--keep", "--no-oracle", "--repeat", "2"], capture_output=True, text=True)
    if e2e.returncode != 0:
        print(e2e.stdout[-2000:], e2e.stderr[-2000:])
        raise SystemExit(1)
    e2e_res = json.loads(e2e.stdout.strip().split("\n")[-1])
    raw = open(os.path.join(args.workdir, "prep", "portcullis.sorted.alignments.bam"), "rb").read()
    raw = raw * args.times
    offs = block_offsets(raw)
    from portcullis_amd import ffi
    ctx = ffi.Context(0, "UNKNOWN", flags=ffi.FLAG_KERNEL_TIMING)
    chunk = args.chunk_mb << 20
    cuts = [0]
    for o in offs:
        if o - cuts[-1] >= chunk:
            cuts.append(o)
    if cuts[-1] != len(raw):
        cuts.append(len(raw))
    # warm-up on the first chunk (allocations), then timed passes
    ctx.inflate_bgzf(raw[cuts[0]:cuts[1]])
    ctx.reset_kernel_timing()
    total_out = 0
    t0 = time.perf_counter()
    first = None
    for a, b in zip(cuts[:-1], cuts[1:]):
        out = ctx.inflate_bgzf(raw[a:b])
        total_out += len(out)
        if first is None:
            first = (raw[a:b], out)
    wall = time.perf_counter() - t0
    kt_all = ctx.kernel_timing()
    parts = {k: v for k, v in kt_all.items() if k in ("bgzf_inflate", "bgzf_decode", "bgzf_resolve")}
    kt = (max(v[0] for v in parts.values()), sum(v[1] for v in parts.values()))
    # zlib on a sample of blocks (single thread)
    sample_in, sample_out = first
    so = block_offsets(sample_in)
    n_s = min(len(so) - 1, 4000)
    t = time.perf_counter()
    got = 0
    for i in range(n_s):
        blk = sample_in[so[i]:so[i + 1]]
        got += len(zlib.decompress(blk[18:-8], -15))
    dt = time.perf_counter() - t
    want = b"".join(zlib.decompress(sample_in[so[i]:so[i + 1]][18:-8], -15) for i in range(n_s))
    assert bytes(sample_out[:len(want)]) == want, "device inflate differs from zlib on the sampled blocks"
    print(json.dumps({
        "bam_mb": round(len(raw) / 1e6, 1), "inflated_mb": round(total_out / 1e6, 1), "blocks": len(offs) - 1,
        "kernel_ms": round(kt[1], 2), "kernel_launches": kt[0],
        "kernels": {k: {"launches": v[0], "ms": round(v[1], 3)} for k, v in parts.items()},
        "zlib_checked_blocks": n_s,
        "kernel_gbps_inflated": round(total_out / (kt[1] * 1e-3) / 1e9, 2),
        "kernel_gbps_compressed": round(len(raw) / (kt[1] * 1e-3) / 1e9, 2),
        "host_to_host_wall_s": round(wall, 3),
        "zlib_1thread_gbps_inflated": round(got / dt / 1e9, 3),
        "e2e": e2e_res,
    }))
    ctx.close()


if __name__ == "__main__":
    main()
