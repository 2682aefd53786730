#!/usr/bin/env python3
"""Device-resident run of the BASELINE.json configs[2] shape on one MI355X: 200 M paired-end 150-bp
reads over 25 contigs with human-like lengths (GRCh38 chr1 248 Mb ... chrM 16.6 kb, 3.1 Gb in total;
reads and ~250 k junctions spread in proportion to contig length), orientation FR.

Not the driver's bench (bench.py measures configs[1], the configuration the metric is quoted on);
this shows the same hot path at the 200 M-read scale, every contig's records resident in HBM.
One step = submit + finish every contig on one context + collect the merged row table.

    python tests/bench_c3.py [--reads 200000000] [--junctions 250000] [--steps 3] [--check-contigs 1]

Prints one JSON line.  --check-contigs K re-runs the first K contigs through the CPU oracle and
compares the device rows (bit-exact integers, entropy 1e-6).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repository root (this file lives in tests/: it checks against the oracle)
# chr1..22, X, Y, M
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
          135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
          46709983, 50818468, 156040895, 57227415, 16569]
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200_000_000, help="total over all contigs")
    ap.add_argument("--junctions", type=int, default=250_000, help="total over all contigs")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--check-contigs", type=int, default=1)
    args = ap.parse_args()

    import numpy as np
    import torch

    from portcullis_amd import ffi, synth

    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X (no CPU fallback)")
    dev = torch.device("cuda", 0)
    ctx = ffi.Context(device=0, orientation="FR")
    lens = GRCH38
    tot_len = sum(lens)
    ctx.set_refs(lens)
    cfgs, batches, genomes_host = [], [], {}
    n_reads = n_pairs = 0
    t_gen = time.time()
    for i, ln in enumerate(lens):
        cfg = synth.SynthConfig(f"C3-{i}", ln, max(200, round(args.reads * ln / tot_len)),
                                max(2, round(args.junctions * ln / tot_len)), 150, paired=True, seed=77_000 + i)
        d = synth.generate(cfg, device=dev, tid=i)
        ctx.upload_contig_device(i, d["genome"])
        if i < args.check_contigs:
            genomes_host[i] = d["genome"].cpu().numpy().tobytes()
        cfgs.append(cfg)
        batches.append((d["batch"], d["n_reads"], d["n_pairs"]))
        n_reads += d["n_reads"]
        n_pairs += d["n_pairs"]
        del d
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    hbm_gb = torch.cuda.memory_allocated() / 1e9

    state = {}

    def step():
        ctx.clear_rows()
        regs = []
        for i, (b, n, _) in enumerate(batches):
            ctx.submit_batch_device(i, b, n)
            regs.append(ctx.finish_contig(i))
        state["regs"] = regs
        state["rows"] = ctx.collect(copy=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rows = state["rows"].copy()
    regs = state["regs"]
    assert sum(r["n_reads"] for r in regs) == n_reads and sum(r["n_pairs"] for r in regs) == n_pairs
    assert int(rows["nb_raw"].astype(np.int64).sum()) == n_pairs  # every N op lands in exactly one junction
    key = (rows["refid"].astype(np.int64) << 40) | (rows["start"].astype(np.int64) << 8)
    assert (np.diff(key) >= 0).all()  # contig-major, sorted by start inside a contig

    checked = None
    if args.check_contigs > 0:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from oracle import oracle as orc
        from parity import assert_rows_equal, region_equal

        t_cpu = 0.0
        n_cpu = 0
        worst = 0.0
        for i in range(min(args.check_contigs, len(lens))):
            b, n, _ = batches[i]
            hb = synth.batch_to_numpy(b, 0, n)
            t = time.perf_counter()
            orows, oreg = orc.find_juncs(i, lens[i], genomes_host[i], hb.to_oracle(), "FR")
            t_cpu += time.perf_counter() - t
            n_cpu += n
            region_equal(regs[i], oreg)
            worst = max(worst, assert_rows_equal(rows[rows["refid"] == i], orows))
        checked = {"contigs": min(args.check_contigs, len(lens)), "oracle_reads_per_sec": n_cpu / t_cpu, "cores": 1,
                   "max_entropy_diff": worst}

    print(json.dumps({
        "workload": f"BASELINE configs[2] shape: {n_reads} paired-end 150-bp reads, {len(lens)} contigs of GRCh38 lengths "
                    f"({tot_len} bp), {len(rows)} junctions, {n_pairs} spliced pairs; device-resident records, orientation FR",
        "reads_per_sec": n_reads * args.steps / elapsed,
        "junctions_per_sec": len(rows) * args.steps / elapsed,
        "ms_per_step": elapsed / args.steps * 1e3,
        "steps": args.steps,
        "n_gpus": 1,
        "hbm_resident_gb": round(hbm_gb, 2),
        "datagen_s": round(t_gen, 1),
        "oracle_check": checked,
    }))
    ctx.close()


if __name__ == "__main__":
    main()
