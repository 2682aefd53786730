#!/usr/bin/env python3
"""A longer differential run than the test suite affords: many seeds, generator options and orientations, the HIP
path (SoA batches, split batches and BAM bytes through pjb_submit_bam) against the oracle.  Run under gpurun:

    python tests/fuzz_campaign.py --seeds 200 [--start 1000]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=100)
    ap.add_argument("--start", type=int, default=1000)
    ap.add_argument("--max-reads", type=int, default=6000)
    args = ap.parse_args()
    from fuzzgen import EXC_OPTS, make_reads, to_batch
    from oracle import oracle as orc
    from parity import assert_rows_equal, region_equal
    from portcullis_amd import ffi
    from test_gpu_ingest import bam_targets
    from util_bam import write_bam

    oris = ["UNKNOWN", "FR", "RF", "FF", "SE"]
    # (the last two: round 6's "exception channel" family -- a genome of pure ACGT but for letters planted at the anchors' edges, reads with
    # N / IUPAC codes / '=' of their own -- alone and with many indels: the 2-bit and the 4-bit compare side by side in one wavefront)
    opt_sets = [None, dict(indel=0.3, clip=0.4, hard=0.1), dict(eqx=0.5, pad=0.1, sub=0.05), dict(indel=0.0, clip=0.0, hard=0.0, sub=0.0),
                dict(EXC_OPTS), dict(EXC_OPTS, indel=0.3, clip=0.4)]
    t0 = time.time()
    n_j = 0
    fails = []
    for k in range(args.seeds):
        seed = args.start + k
        rng = np.random.default_rng(seed)
        ori = oris[seed % len(oris)]
        opts = opt_sets[(seed // 5) % len(opt_sets)]
        n_reads = int(rng.integers(200, args.max_reads))
        glen = int(rng.integers(8000, 60000))
        L = (int(rng.integers(20, 60)), int(rng.integers(60, 260)))
        genome, reads = make_reads(seed, glen=glen, n_reads=n_reads, paired=(seed % 2 == 1), opts=opts, L=L, n_tx=int(rng.integers(2, 30)))
        batch = to_batch(reads)
        try:
            orows, oreg = orc.find_juncs(0, len(genome), genome, batch, ori)
            with ffi.Context(0, ori) as ctx:
                ctx.set_refs([len(genome), 1000])
                # 1. one batch
                drows, dreg = ffi.run_contig(ctx, 0, genome.encode(), [batch])
                region_equal(dreg, oreg)
                assert_rows_equal(drows, orows)
                # 1b. the same batch with 4-bit bases only (pjb_batch.seq2 = NULL: round 5's compare)
                if seed % 3 == 0:
                    ctx.clear_rows()
                    ctx.submit_batch(0, batch, seq2=False)
                    dreg = ctx.finish_contig(0)
                    region_equal(dreg, oreg)
                    assert_rows_equal(ctx.collect(), orows)
                # 2. ragged batches
                cuts = sorted(set(int(x) for x in rng.integers(1, max(2, batch.n), size=int(rng.integers(1, 6)))))
                cuts = [0] + [c for c in cuts if 0 < c < batch.n] + [batch.n]
                ctx.clear_rows()
                for a, b in zip(cuts[:-1], cuts[1:]):
                    ctx.submit_batch(0, batch.slice(a, b))
                dreg = ctx.finish_contig(0)
                region_equal(dreg, oreg)
                assert_rows_equal(ctx.collect(), orows)
                # 3. BAM bytes
                with tempfile.TemporaryDirectory() as d:
                    path = os.path.join(d, "f.bam")
                    for i, r in enumerate(reads):
                        r["tid"] = 0
                        r["name"] = f"s{seed}r{i}"
                    write_bam(path, [("c", len(genome)), ("d", 1000)], reads, block_size=int(rng.integers(300, 0xFF00)), write_index=False,
                              level=int(rng.integers(0, 10)))
                    raw, _, first = bam_targets(path)
                if 0 in first:
                    coff, uoff = first[0]
                    ctx.clear_rows()
                    n = ctx.submit_bam(0, raw[coff:], uoff)
                    assert n == batch.n
                    dreg = ctx.finish_contig(0)
                    region_equal(dreg, oreg)
                    assert_rows_equal(ctx.collect(), orows)
            n_j += len(orows)
        except Exception as e:  # keep going: report every failing seed
            fails.append((seed, ori, repr(e)[:300]))
    print(f"{args.seeds} seeds from {args.start}: {n_j} junctions compared three ways each (a third of the seeds a fourth way: 4-bit bases only), {len(fails)} failures, {time.time() - t0:.0f} s")
    for f in fails[:20]:
        print("FAIL", f)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
