#!/usr/bin/env python3
"""Measurement of `junc --extra`: two targets of the BASELINE configs[1] shape through a
PJB_FLAG_EXTRA context (per-target depth of the unspliced records, flanking alignment counts, name codes; then
pjb_extra_finish: multiple-mapping score and coverage of every junction), against the same targets without the flag.
Parity with the oracle is checked on a 2 x 2 M-read copy of the workload (the oracle's pileup is per base and per
record); the timing runs on 2 x 10 M reads.  Prints one JSON line."""
import dataclasses
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make(cfg, synth, torch, seeds=(1, 2)):
    out = []
    for tid, seed in enumerate(seeds):
        d = synth.generate(dataclasses.replace(cfg, seed=seed), device="cuda", tid=tid)
        g = torch.Generator(device="cuda")
        g.manual_seed(100 + seed)
        n = d["n_reads"]
        # a fifth of the records share their name code with another record (multi-mapped fragments, also across targets)
        codes = torch.randint(1, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
        dup = torch.rand(n, device="cuda", generator=g) < 0.2
        codes = torch.where(dup, codes % 50021 + 7, codes)
        d["batch"]["name_hash"] = codes
        out.append(d)
    torch.cuda.synchronize()
    return out


def run_device(ffi, torch, data, lens, extra, reps):
    flags = ffi.FLAG_EXTRA if extra else 0
    best = None
    with ffi.Context(0, "UNKNOWN", flags=flags) as ctx:
        ctx.set_refs(lens)
        for tid, d in enumerate(data):
            ctx.upload_contig_device(tid, d["genome"])
        for _ in range(reps + 1):
            ctx.clear_rows()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for tid, d in enumerate(data):   # both targets' chains queued: they run side by side (with and without the flag)
                ctx.submit_batch_device(tid, d["batch"], d["n_reads"])
                ctx.finish_contig_begin(tid)
            for tid in range(len(data)):
                ctx.finish_contig_end(tid)
            rows = ctx.collect(copy=False)  # (waits for the rows' DMA; the table stays in the context's page-locked memory)
            xr = ctx.extra_finish(copy=False) if extra else None
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            rows = rows.copy()
            xr = xr.copy() if extra else None
    return rows, xr, best


def main():
    import numpy as np
    import torch

    from oracle import oracle as orc
    from portcullis_amd import ffi, synth

    base = synth.CONFIGS["C2"]
    # ---- parity on the small copy
    small = dataclasses.replace(base, n_reads=2_000_000, contig_len=20_000_000, n_junctions=10_000)
    data = make(small, synth, torch)
    lens = [small.contig_len] * 2
    rows, xr, _ = run_device(ffi, torch, data, lens, True, 0)
    soa, nh, rows_all = {}, {}, []
    tot = sum_len = max_len = 0
    t0 = time.perf_counter()
    for tid, d in enumerate(data):
        hb = synth.batch_to_numpy(d["batch"], 0, d["n_reads"])
        hb.name_hash = d["batch"]["name_hash"].cpu().numpy().view(np.uint64)
        soa[tid], nh[tid] = hb, hb.name_hash
        r, reg = orc.find_juncs(tid, lens[tid], d["genome"].cpu().numpy().tobytes().decode(), hb, "UNKNOWN")
        rows_all.append(r)
        tot += reg["spliced"] + reg["unspliced"]
        sum_len += reg["sum_len"]
        max_len = max(max_len, reg["max_len"])
    t_junc = time.perf_counter() - t0
    orows = orc.finalize(np.concatenate(rows_all), sum_len / tot)
    t0 = time.perf_counter()
    orows = orc.extra(lens, soa, nh, orows, max_len)
    t_extra_cpu = time.perf_counter() - t0
    assert len(rows) == len(orows) == len(xr)
    for col in ("coverage", "up_aln", "down_aln"):
        assert (xr[col] == orows[col]).all(), col
    worst = float(np.abs(xr["mm_score"] - orows["mm_score"]).max())
    assert worst <= 1e-6
    n_small = sum(d["n_reads"] for d in data)
    del data
    # ---- timing on 2 x 10 M reads
    data = make(base, synth, torch)
    lens = [base.contig_len] * 2
    n = sum(d["n_reads"] for d in data)
    rows0, _, t_plain = run_device(ffi, torch, data, lens, False, 3)
    rows1, xr1, t_extra = run_device(ffi, torch, data, lens, True, 3)
    assert rows0.tobytes() == rows1.tobytes()
    print(json.dumps({"workload": f"junc --extra, 2 targets of the BASELINE configs[1] shape: {n} alignments, {len(rows1)} junctions, "
                                  "a fifth of the records multi-mapped",
                      "plain_ms": round(t_plain * 1e3, 2), "extra_ms": round(t_extra * 1e3, 2),
                      "reads_per_sec_plain": n / t_plain, "reads_per_sec_extra": n / t_extra,
                      "note": "both targets queued (pjb_finish_contig_begin / _end), device-resident records, rows and extra rows on the host",
                      "extra_over_plain": round(t_extra / t_plain, 2),
                      "cpu_oracle": {"sample": f"{n_small} alignments (2 x 2 M reads), every coverage / up_aln / down_aln equal, max |mm_score diff| "
                                               f"{worst:.2g}", "junc_s": round(t_junc, 2), "extra_s": round(t_extra_cpu, 2),
                                     "reads_per_sec_extra": n_small / (t_junc + t_extra_cpu), "cores": 1}}))


if __name__ == "__main__":
    main()
