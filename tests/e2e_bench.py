#!/usr/bin/env python3
"""End-to-end `junc` run on a synthetic prepared BAM: BGZF/BAM decode on host threads -> SoA batches
-> H2D -> HIP pipeline -> .tab/.bed, through the portcullis_amd program, checked byte for byte
against the CPU oracle's .tab for the same records.

    python tests/e2e_bench.py --config C2 --threads 8 [--workdir /tmp/e2e] [--keep]

Prints one JSON line with wall-clock reads/s.  Reference points for the same stage (SURVEY.md
section 6, measured with the real reference during the survey): 0.37-0.39 M reads/s per thread,
one thread per contig at most."""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repository root (this file lives in tests/: it checks against the oracle)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def dump_contig(d, name, data):
    import numpy as np
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "name.txt"), "w").write(name)
    data["genome"].cpu().numpy().tofile(os.path.join(d, "genome.u8"))
    b = data["batch"]
    ext = dict(pos="i32", flag="u16", mapq="u8", xs="u8", l_qseq="i32", mtid="i32", mpos="i32", cig_off="u32",
               cigar="u32", seq_off="u32", seq4="u8")
    for k, e in ext.items():
        b[k].cpu().numpy().tofile(os.path.join(d, f"{k}.{e}"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--workdir", default="/tmp/pjb_e2e")
    ap.add_argument("--contigs", type=int, default=1)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--orientation", default="UNKNOWN")
    ap.add_argument("--strandedness", default="UNKNOWN")
    ap.add_argument("--scale-contigs", action="store_true", help="contig k gets 1/(k+1) of the reads and length")
    args = ap.parse_args()
    import torch
    from portcullis_amd import synth

    cfg = synth.CONFIGS[args.config]
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    wd = args.workdir
    shutil.rmtree(wd, ignore_errors=True)
    prep = os.path.join(wd, "prep")
    os.makedirs(prep)
    t0 = time.time()
    datas, dirs = [], []
    import dataclasses
    cfgs = []
    for c in range(args.contigs):
        cc = cfg
        if args.scale_contigs and c > 0:
            f = 1.0 / (c + 1)
            cc = dataclasses.replace(cfg, contig_len=max(20_000, int(cfg.contig_len * f)), n_reads=max(500, int(cfg.n_reads * f)),
                                     n_junctions=max(8, int(cfg.n_junctions * f)))
        cfgs.append(cc)
        d = synth.generate(cc, device=dev, seed=cfg.seed + c, tid=c)
        datas.append(d)
        cd = os.path.join(wd, f"contig{c}")
        dump_contig(cd, f"chr{c + 1}", d)
        dirs.append(cd)
    exe = os.path.join(ROOT, "tools", "soa2bam")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "soa2bam.cc"), "-lz", "-lpthread"])
    subprocess.check_call([exe, prep, str(args.threads)] + dirs)
    t_prep = time.time() - t0
    bam_bytes = os.path.getsize(os.path.join(prep, "portcullis.sorted.alignments.bam"))
    n_reads = sum(d["n_reads"] for d in datas)
    cli = os.path.join(ROOT, "portcullis_amd", "host", "portcullis_amd")
    walls = []
    for rep in range(args.repeat):
        out = os.path.join(wd, f"out{rep}", "pc")
        t = time.time()
        p = subprocess.run([cli, "junc", "-t", str(args.threads), "--orientation", args.orientation, "--strandedness",
                            args.strandedness, "-o", out, prep], capture_output=True, text=True)
        walls.append(time.time() - t)
        if p.returncode != 0:
            print(p.stdout[-3000:], p.stderr[-3000:])
            raise SystemExit("portcullis_amd failed")
    tab = open(out + ".junctions.tab", "rb").read()
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 1)
    except Exception:
        pass
    res = dict(cpu_quota_cores=quota, config=cfg.name, contigs=args.contigs, reads=n_reads, threads=args.threads, bam_mb=round(bam_bytes / 1e6, 1),
               wall_s=[round(w, 3) for w in walls], reads_per_s=n_reads / min(walls), tab_md5=hashlib.md5(tab).hexdigest(),
               junctions=tab.count(b"\n") - 2, prep_s=round(t_prep, 1), host_cores=os.cpu_count())
    if not args.no_oracle:
        from oracle import oracle as orc
        refs = [(f"chr{c + 1}", cfgs[c].contig_len) for c in range(args.contigs)]
        genomes = {c: datas[c]["genome"].cpu().numpy().tobytes() for c in range(args.contigs)}
        batches = {c: synth.batch_to_numpy(datas[c]["batch"]) for c in range(args.contigs)}
        t = time.time()
        rows, tot = orc.run_prep_like(refs, genomes, batches, args.orientation)
        res["oracle_s"] = round(time.time() - t, 2)
        otab = orc.write_tab(rows, [n for n, _ in refs], [l for _, l in refs])
        res["tab_identical_to_oracle"] = otab == tab
        if otab != tab:
            a, b = tab.split(b"\n"), otab.split(b"\n")
            for i, (x, y) in enumerate(zip(a, b)):
                if x != y:
                    res["first_diff_line"] = i
                    res["got"] = x.decode(errors="replace")[:300]
                    res["exp"] = y.decode(errors="replace")[:300]
                    break
    print(json.dumps(res))
    if not args.keep:
        shutil.rmtree(wd, ignore_errors=True)
    if not args.no_oracle and not res["tab_identical_to_oracle"]:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
