#!/usr/bin/env python3
"""`portcullis_amd bamfilt` end to end on the BASELINE configs[1] BAM (10 M reads, 1.1 GB; run under gpurun): the prepared
directory comes from tests/e2e_bench.py, `junc` writes the table, two thirds of its junctions "pass", and the program filters
the BAM.  Prints one JSON line: wall seconds per run, alignments/s, the md5 of the kept records (inflated bytes) so that two
builds can be compared, and what `zlib` alone needs to inflate the input / deflate the output on one core.

    python tools/bench_bamfilt_program.py [--threads 16] [--runs 3] [--config C2]
"""
import argparse
import gzip
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--workdir", default="/tmp/pjb_bamfilt")
    ap.add_argument("--exe", default=os.path.join(ROOT, "portcullis_amd", "host", "portcullis_amd"), help="the program to time")
    ap.add_argument("--env", action="append", default=[], help="NAME=VALUE for the program (repeatable)")
    args = ap.parse_args()
    wd = args.workdir
    prep = os.path.join(wd, "prep")
    bam = os.path.join(prep, "portcullis.sorted.alignments.bam")
    tab = os.path.join(wd, "out1", "pc.junctions.tab")
    if not (os.path.exists(bam) and os.path.exists(tab)):
        e2e = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "e2e_bench.py"), "--config", args.config, "--threads", str(args.threads),
                              "--workdir", wd, "--keep", "--no-oracle", "--repeat", "2"], capture_output=True, text=True)
        if e2e.returncode != 0:
            print(e2e.stdout[-2000:], e2e.stderr[-2000:])
            raise SystemExit(1)
    lines = open(tab).read().split("\n")
    body = [l for l in lines[1:] if l.strip()]
    kept = [l for k, l in enumerate(body) if k % 3 != 1]
    passed = os.path.join(wd, "pass.junctions.tab")
    open(passed, "w").write("\n".join([lines[0]] + kept) + "\n\n")
    exe = args.exe
    out = os.path.join(wd, "filt", "filtered.bam")
    env = dict(os.environ)
    for kv in args.env:
        k, v = kv.split("=", 1)
        env[k] = v
    walls, summary = [], ""
    for _ in range(args.runs):
        for f in (out, out + ".bai"):   # (a first run: truncating the last run's 1 GB output costs fopen() 90 ms)
            if os.path.exists(f):
                os.remove(f)
        t = time.time()
        p = subprocess.run([exe, "bamfilt", "-o", out, "-c", "HARD", "-t", str(args.threads), passed, bam], capture_output=True, text=True, env=env)
        walls.append(time.time() - t)
        if p.returncode != 0:
            print(p.stdout[-2000:], p.stderr[-2000:])
            raise SystemExit("bamfilt failed")
        summary = [l for l in p.stdout.split("\n") if l.startswith("Filtered out")][0]
    n_in = int(summary.split("In: ")[1].split(";")[0])
    t = time.time()
    h = hashlib.md5()
    n_out_bytes = 0
    with gzip.open(out, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
            n_out_bytes += len(b)
    t_inflate_out = time.time() - t
    print(json.dumps({"workload": f"portcullis_amd bamfilt -c HARD -t {args.threads}, BASELINE configs[1] BAM: {n_in} alignments, "
                                  f"{len(kept)} of {len(body)} junctions pass",
                      "bam_mb": round(os.path.getsize(bam) / 1e6, 1), "out_mb": round(os.path.getsize(out) / 1e6, 1),
                      "out_inflated_mb": round(n_out_bytes / 1e6, 1), "wall_s": [round(w, 3) for w in walls],
                      "alignments_per_sec": n_in / sorted(walls)[len(walls) // 2], "summary": summary, "kept_bytes_md5": h.hexdigest(),
                      "python_gzip_inflate_of_output_s": round(t_inflate_out, 2), "env": args.env, "host_cores": os.cpu_count()}))


if __name__ == "__main__":
    main()
