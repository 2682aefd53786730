#!/usr/bin/env python3
"""Measurement of the `filt` feature rows on the junctions of the BASELINE configs[1] records
(10 M single-end reads, one contig, ~49 k junctions): ModelFeatures::setRow for every junction through
pjb_filt_features (rows and the eight Markov tables in from the host, 34 doubles per junction out: PCIe included), the
kernel's own time (HIP events), and the CPU oracle over the same rows (training excluded) with every value compared."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch

    from oracle import oracle as orc
    from portcullis_amd import ffi, synth

    cfg = synth.CONFIGS["C2"]
    d = synth.generate(cfg, device="cuda")
    torch.cuda.synchronize()
    hb = synth.batch_to_numpy(d["batch"], 0, d["n_reads"])
    genome = d["genome"].cpu().numpy().tobytes().decode()
    t0 = time.perf_counter()
    orows, oreg = orc.find_juncs(0, cfg.contig_len, genome, hb, "UNKNOWN")
    t_junc_cpu = time.perf_counter() - t0
    mean = oreg["sum_len"] / (oreg["spliced"] + oreg["unspliced"])
    orows = orc.finalize(orows, mean)
    n = len(orows)
    idx = np.arange(n)
    good, bad = idx[orows["nb_raw"] >= 3], idx[orows["nb_raw"] < 3]
    sizes = orows["end"] - orows["start"] + 1
    small = idx[sizes <= np.median(sizes)]
    t0 = time.perf_counter()
    G, models, l95 = orc.filt_features([cfg.contig_len], {0: genome}, orows, small, good, good, bad)
    t_cpu_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.filt_features([cfg.contig_len], {0: genome}, orows[:1], small, good, good, bad)  # training + one row
    t_cpu_train = time.perf_counter() - t0
    t_cpu_rows = max(t_cpu_all - t_cpu_train, 1e-9)
    with ffi.Context(0, "UNKNOWN", flags=ffi.FLAG_KERNEL_TIMING) as ctx:
        ctx.set_refs([cfg.contig_len])
        ctx.upload_contig_device(0, d["genome"])
        ctx.submit_batch_device(0, d["batch"], d["n_reads"])
        ctx.finish_contig(0)
        drows = ctx.collect()
        assert len(drows) == n and (drows["start"] == orows["start"]).all()
        F = ctx.filt_features(drows, float(mean), l95, models)
        ctx.reset_kernel_timing()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            F = ctx.filt_features(drows, float(mean), l95, models)
        dt = (time.perf_counter() - t0) / reps
        kt = ctx.kernel_timing()["kg_features"]
    fin = np.isfinite(G)
    assert (np.isnan(F) == np.isnan(G)).all() and (np.isinf(F) == np.isinf(G)).all()
    diff = np.where(fin, np.abs(np.where(fin, F, 0) - np.where(fin, G, 0)), 0.0)
    tol = 1e-6 * np.maximum(1.0, np.abs(np.where(fin, G, 0)))
    assert (diff <= tol).all(), float(diff.max())
    kms = kt[1] / kt[0]
    # per junction: the 200-B row, ~ (4 x 81 + 24 + 23 + 2 x 10) genome bases, 8 + 2 x (77 + 19 + 18) table lookups of 8 B, 272 B out
    alg = n * (200 + 391 + 8 * 236 + 272)
    print(json.dumps({"workload": f"filt feature rows of the BASELINE configs[1] junctions: {n} junctions x {ffi.N_FEATURES} features, "
                                  f"k-mer models of order 5 trained on {len(good)} / {len(bad)} junctions",
                      "junctions_per_sec_incl_pcie": n / dt, "call_ms": round(dt * 1e3, 3), "kernel_ms": round(kms, 4),
                      "alg_bytes": alg, "kernel_gbps": round(alg / (kms * 1e-3) / 1e9, 1),
                      "max_abs_diff_vs_oracle": float(diff.max()),
                      "cpu_oracle": {"junctions_per_sec": n / t_cpu_rows, "cores": 1, "rows_s": round(t_cpu_rows, 3),
                                     "training_s": round(t_cpu_train, 3), "junc_oracle_s": round(t_junc_cpu, 2),
                                     "sample": "all rows (training of the models timed separately and excluded)"}}))


if __name__ == "__main__":
    main()
