#!/usr/bin/env python3
"""Long-read datapoint (run under gpurun): synthetic ~10 kb alignments with ~95 CIGAR operations each (M / I / D runs inside
16 exons, 15 N operations) through the junc chain -- rows compared with the oracle on a subset, then the chain timed on the
whole set, next to 100-bp single-end reads (synth C2) on the same context.  What it answers: does the per-base rate of the
thread-per-read CIGAR walks (k1_count, k1_emit, k4b_generic) hold up when a read has a hundred operations?

    python tools/bench_long_reads.py [--reads 200000] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OP_M, OP_I, OP_D, OP_N = 0, 1, 2, 3
EXONS, EXON_LEN, INTRON_LEN = 16, 600, 1000


def make_long_reads(n_reads, n_genes, glen, seed=7):
    from portcullis_amd.records import ReadBatch

    rng = np.random.default_rng(seed)
    span = EXONS * EXON_LEN + (EXONS - 1) * INTRON_LEN
    starts = np.sort(rng.choice((glen - span - 1000) // 64, size=n_genes, replace=False).astype(np.int64) * 64 + 500)
    gene = np.sort(rng.integers(0, n_genes, size=n_reads))
    pos = starts[gene].astype(np.int32)
    # per exon: M a, D d, M b, I i, M c with a + d + b + c = EXON_LEN
    a = rng.integers(50, 250, size=(n_reads, EXONS))
    d = rng.integers(1, 5, size=(n_reads, EXONS))
    b = rng.integers(50, 200, size=(n_reads, EXONS))
    i = rng.integers(1, 4, size=(n_reads, EXONS))
    c = EXON_LEN - a - d - b
    ops = np.zeros((n_reads, EXONS * 6 - 1), np.uint32)
    for e in range(EXONS):
        o = e * 6
        ops[:, o + 0] = (a[:, e] << 4) | OP_M
        ops[:, o + 1] = (d[:, e] << 4) | OP_D
        ops[:, o + 2] = (b[:, e] << 4) | OP_M
        ops[:, o + 3] = (i[:, e] << 4) | OP_I
        ops[:, o + 4] = (c[:, e] << 4) | OP_M
        if e + 1 < EXONS:
            ops[:, o + 5] = (INTRON_LEN << 4) | OP_N
    n_ops = ops.shape[1]
    lq = (a + b + c + i).sum(axis=1).astype(np.int32)
    words = ((lq + 1) // 2 + 3) // 4
    seq_off = np.zeros(n_reads + 1, np.uint32)
    seq_off[1:] = np.cumsum(words)
    seq4 = rng.integers(0, 256, size=int(seq_off[-1]) * 4, dtype=np.uint8)
    seq4 = ((seq4 & 0x33) + 0x11) & 0xFF  # nibbles in {1, 2, 4} ... keep them letters of ACGT: map below
    # nibble values 1/2/3/4 -> codes of A C M G; force to A C G T codes (1, 2, 4, 8)
    lut = np.array([1, 1, 2, 4, 8] + [1] * 11, np.uint8)
    seq4 = (lut[seq4 >> 4] << 4) | lut[seq4 & 15]
    cig_off = (np.arange(n_reads + 1, dtype=np.uint64) * n_ops).astype(np.uint32)
    return ReadBatch(pos, np.zeros(n_reads, np.uint16), np.full(n_reads, 60, np.uint8), np.ones(n_reads, np.uint8), lq,
                     np.full(n_reads, -1, np.int32), np.full(n_reads, -1, np.int32), cig_off, ops.reshape(-1), seq_off, seq4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200_000)
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from oracle import oracle as orc
    from parity import assert_rows_equal, region_equal
    from portcullis_amd import ffi, synth

    glen = 200_000_000
    rng = np.random.default_rng(1)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=glen).tobytes()
    batch = make_long_reads(args.reads, args.genes, glen)
    bases = int(batch.l_qseq.astype(np.int64).sum())
    out = {"long": {"reads": batch.n, "bases": bases, "ops_per_read": int(batch.cig_off[1]), "mean_len": bases / batch.n}}
    with ffi.Context(0, "UNKNOWN", flags=ffi.FLAG_KERNEL_TIMING) as ctx:
        ctx.set_refs([glen, 100_000_000])
        ctx.upload_contig(0, genome)
        # parity on a subset
        sub = batch.slice(0, min(batch.n, 3000))
        orows, oreg = orc.find_juncs(0, glen, genome, sub, "UNKNOWN")
        ctx.clear_rows()
        ctx.submit_batch(0, sub)
        region_equal(ctx.finish_contig(0), oreg)
        out["long"]["subset_max_entropy_diff"] = assert_rows_equal(ctx.collect(), orows)
        out["long"]["subset_junctions"] = len(orows)
        # timing: records resident in HBM
        dev = {k: torch.from_numpy(np.ascontiguousarray(getattr(batch, k))).cuda() for k in
               ("pos", "flag", "mapq", "xs", "l_qseq", "mtid", "mpos", "cig_off", "cigar", "seq_off", "seq4")}
        for _ in range(2):
            ctx.clear_rows()
            ctx.submit_batch_device(0, dev, batch.n)
            reg = ctx.finish_contig(0)
        ctx.reset_kernel_timing()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.clear_rows()
            ctx.submit_batch_device(0, dev, batch.n)
            reg = ctx.finish_contig(0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        kt = ctx.kernel_timing()
        out["long"].update(ms_per_step=dt * 1e3, reads_per_s=batch.n / dt, bases_per_s=bases / dt, pairs=reg["n_pairs"], junctions=reg["n_junctions"],
                           kernels_ms={k: round(v[1] / args.steps, 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1])[:8]})
        # the same context on 100-bp single-end reads (BASELINE configs[1])
        d = synth.generate(synth.CONFIGS["C2"], device="cuda", tid=1)
        ctx.upload_contig_device(1, d["genome"])
        for _ in range(2):
            ctx.clear_rows()
            ctx.submit_batch_device(1, d["batch"], d["n_reads"])
            ctx.finish_contig(1)
        ctx.reset_kernel_timing()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.clear_rows()
            ctx.submit_batch_device(1, d["batch"], d["n_reads"])
            reg = ctx.finish_contig(1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        kt = ctx.kernel_timing()
        sb = d["n_reads"] * synth.CONFIGS["C2"].read_len
        out["short"] = dict(reads=d["n_reads"], bases=sb, ms_per_step=dt * 1e3, reads_per_s=d["n_reads"] / dt, bases_per_s=sb / dt, pairs=reg["n_pairs"],
                            kernels_ms={k: round(v[1] / args.steps, 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1])[:8]})
    out["per_base_rate_long_over_short"] = out["long"]["bases_per_s"] / out["short"]["bases_per_s"]
    out["per_pair_us"] = {"long": out["long"]["ms_per_step"] * 1e3 / out["long"]["pairs"], "short": out["short"]["ms_per_step"] * 1e3 / out["short"]["pairs"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
