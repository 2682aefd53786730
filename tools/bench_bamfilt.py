#!/usr/bin/env python3
"""Measurement of the `bamfilt` decision kernel on the BASELINE configs[1] records: 10 M single-end
reads, one contig, two thirds of the junctions passing.  Prints one JSON line: alignments/s through pjb_filter_batch
(host arrays in, codes out: PCIe included), the kernel's own time (HIP events) against its algorithmic bytes
(pos 4 + cig_off 4 + 4 per CIGAR op + 1 code per alignment, + 8 per probed key), and the CPU oracle on a 1 M prefix
with the codes compared."""
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
    with ffi.Context(0, "UNKNOWN", flags=ffi.FLAG_KERNEL_TIMING) as ctx:
        ctx.set_refs([cfg.contig_len])
        ctx.upload_contig_device(0, d["genome"])
        ctx.submit_batch_device(0, d["batch"], d["n_reads"])
        ctx.finish_contig(0)
        rows = ctx.collect()
        keep = np.arange(len(rows)) % 3 != 1
        ctx.filter_set_junctions(0, rows["start"][keep], rows["end"][keep])
        ctx.filter_batch(0, hb, "HARD")
        ctx.reset_kernel_timing()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            codes = ctx.filter_batch(0, hb, "HARD")
        dt = (time.perf_counter() - t0) / reps
        kt = ctx.kernel_timing()["kf_filter"]
    n = d["n_reads"]
    alg = n * (4 + 4 + 1) + 4 * d["n_cigar_ops"] + 8 * d["n_pairs"]
    kms = kt[1] / kt[0]
    m = 200_000
    sub = hb.slice(0, m)
    t0 = time.perf_counter()
    want = orc.bamfilt_flags(sub, rows["start"][keep], rows["end"][keep], "HARD")
    t_cpu = time.perf_counter() - t0
    assert (codes[:m] == want).all()
    print(json.dumps({"workload": f"bamfilt decision, BASELINE configs[1] records: {n} alignments, {int(keep.sum())} of {len(rows)} junctions pass",
                      "alignments_per_sec_incl_pcie": n / dt, "kernel_ms": round(kms, 4), "alg_bytes": alg,
                      "kernel_gbps": round(alg / (kms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(alg / (kms * 1e-3) / 8e12, 4),
                      "kept": int((codes > 0).sum()), "dropped": int((codes == 0).sum()), "modified": int((codes == 3).sum()),
                      "cpu_oracle": {"alignments_per_sec": m / t_cpu, "cores": 1, "sample": f"first {m} alignments, codes equal the device's",
                                     "note": "the oracle probes the junction list linearly (literal restatement), not a hash map"}}))


if __name__ == "__main__":
    main()
