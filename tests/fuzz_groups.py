#!/usr/bin/env python3
"""Differential run for target groups: per seed a handful of fuzz targets (generator options, read lengths, orientations as
in fuzz_campaign.py), cut into random groups in a random order, through pjb_finish_group_begin / _end -- rows and
per-target results against the oracle's for every target.  Run under gpurun:

    python tests/fuzz_groups.py --seeds 150 [--start 5000]
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
    ap.add_argument("--seeds", type=int, default=100)
    ap.add_argument("--start", type=int, default=5000)
    ap.add_argument("--max-reads", type=int, default=4000)
    args = ap.parse_args()
    from fuzzgen import make_reads, to_batch
    from oracle import oracle as orc
    from parity import assert_rows_equal, region_equal
    from portcullis_amd import ffi

    oris = ["UNKNOWN", "FR", "RF", "FF", "SE"]
    opt_sets = [None, dict(indel=0.3, clip=0.4, hard=0.1), dict(eqx=0.5, pad=0.1, sub=0.05), dict(indel=0.0, clip=0.0, hard=0.0, sub=0.0)]
    t0 = time.time()
    n_j = n_t = 0
    fails = []
    for k in range(args.seeds):
        seed = args.start + k
        rng = np.random.default_rng(seed)
        ori = oris[seed % len(oris)]
        n_targets = int(rng.integers(2, 8))
        targets = []
        try:
            for tid in range(n_targets):
                if rng.random() < 0.12:
                    targets.append(("ACGT" * int(rng.integers(50, 500)), None, None, None))  # a target without alignments
                    continue
                opts = opt_sets[int(rng.integers(0, len(opt_sets)))]
                L = (int(rng.integers(20, 60)), int(rng.integers(60, 260)))
                genome, reads = make_reads(seed * 16 + tid, glen=int(rng.integers(8000, 60000)), n_reads=int(rng.integers(100, args.max_reads)),
                                           paired=bool(rng.integers(0, 2)), opts=opts, L=L, n_tx=int(rng.integers(2, 30)))
                batch = to_batch(reads)
                orows, oreg = orc.find_juncs(tid, len(genome), genome, batch, ori)
                targets.append((genome, batch, orows, oreg))
            # random groups over a random order of the targets
            order = [int(x) for x in rng.permutation(n_targets)]
            groups = []
            while order:
                n = int(rng.integers(1, len(order) + 1))
                groups.append(order[:n])
                order = order[n:]
            with ffi.Context(0, ori) as ctx:
                ctx.set_refs([len(t[0]) for t in targets])
                for tid, t in enumerate(targets):
                    ctx.upload_contig(tid, t[0].encode())
                ctx.clear_rows()
                regs, queued = {}, []
                for g in groups:
                    for tid in g:
                        b = targets[tid][1]
                        if b is None:
                            continue
                        if b.n > 20 and rng.random() < 0.5:  # ragged batches
                            cut = int(rng.integers(1, b.n))
                            ctx.submit_batch(tid, b.slice(0, cut))
                            ctx.submit_batch(tid, b.slice(cut, b.n))
                        else:
                            ctx.submit_batch(tid, b)
                    ctx.finish_group_begin(g)
                    queued.append(g)
                    if len(queued) >= int(rng.integers(1, ffi.MAX_QUEUED + 1)):
                        regs.update(ctx.finish_group_end(queued.pop(0)))
                while queued:
                    regs.update(ctx.finish_group_end(queued.pop(0)))
                rows = ctx.collect()
            seen = [int(x) for x in dict.fromkeys(rows["refid"].tolist())]
            want_order = [t for g in groups for t in g if targets[t][1] is not None and len(targets[t][2])]
            assert seen == want_order, (seen, want_order)
            for tid, t in enumerate(targets):
                if t[1] is None:
                    assert regs[tid]["n_reads"] == 0
                    continue
                region_equal(regs[tid], t[3])
                assert_rows_equal(rows[rows["refid"] == tid], t[2])
                n_j += len(t[2])
                n_t += 1
        except Exception as e:  # noqa: BLE001
            fails.append((seed, repr(e)[:300]))
            print("FAIL seed", seed, repr(e)[:300], flush=True)
    print(f"{args.seeds} seeds, {n_t} targets, {n_j} junctions, {len(fails)} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
