#!/usr/bin/env python3
"""Differential run of `junc --extra` longer than the test suite affords: many seeds, two to six targets each, chains queued
one to four deep, through the records themselves (the default) and through the depth vector; mm_score, coverage, up_aln,
down_aln of every junction against the oracle.  Run under gpurun:

    python tests/fuzz_extra.py --seeds 60 [--start 7000]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=60)
    ap.add_argument("--start", type=int, default=7000)
    args = ap.parse_args()
    from extra_util import add_names, assert_extra_equal, device_extra, oracle_extra
    from fuzzgen import make_reads
    from oracle import oracle as orc
    from parity import assert_rows_equal
    from portcullis_amd import ffi
    t0 = time.time()
    n_junc = n_targets = failures = 0
    for seed in range(args.start, args.start + args.seeds):
        rng = np.random.default_rng(seed)
        paired = bool(rng.integers(0, 2))
        contigs, pool = [], []
        for t in range(int(rng.integers(2, 7))):
            if rng.random() < 0.1:
                contigs.append(("ACGT" * 300, None))
                continue
            genome, reads = make_reads(seed * 17 + t, n_reads=int(rng.integers(300, 3500)), paired=paired, glen=int(rng.integers(4000, 40000)),
                                       L=(int(rng.integers(25, 60)), int(rng.integers(60, 160))))
            add_names(reads, rng, f"s{seed}c{t}", pool, share=float(rng.uniform(0.0, 0.5)))
            contigs.append((genome, reads))
        orientation = "FR" if paired else "UNKNOWN"
        try:
            orows, _ = oracle_extra(orc, contigs, orientation)
            for dense in (False, True):
                rows, extra = device_extra(ffi, orc, contigs, orientation, queue=int(rng.integers(1, 5)), dense=dense,
                                           split=(0.3, 0.7) if rng.random() < 0.3 else None)
                assert_rows_equal(rows, orows)
                assert_extra_equal(rows, extra, orows)
            n_junc += len(orows)
            n_targets += len(contigs)
        except Exception as e:  # noqa: BLE001
            failures += 1
            print("seed", seed, "FAILED:", repr(e)[:300], flush=True)
    print(f"{args.seeds} seeds, {n_targets} targets, {n_junc} junctions, both paths: {failures} failures, {time.time() - t0:.0f} s")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
