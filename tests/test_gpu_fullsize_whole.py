"""BASELINE configs[4] whole on one GPU -- in a module of its own: the full-size fixtures of test_gpu_fullsize.py (contexts that
hold tens of GB of chain scratch) are closed by the time this one allocates its 120 GB."""
import hashlib
import os

import numpy as np
import pytest

from parity import assert_rows_equal, region_equal

pytestmark = pytest.mark.gpu


# ---- BASELINE configs[4] WHOLE on one GPU: 1 B paired-end reads over the 25 GRCh38-sized targets, 300 k junctions, Zipf depth,
# strandedness=firststrand -- 73 GB of records resident in HBM, finished as three kernel chains (groups of ~335 M reads, 110 M
# pairs each).  Size-independent properties on the whole table; chr1 (80 M reads) and the target of the deepest junction
# against the oracle.
@pytest.mark.skipif(os.environ.get("PJB_SKIP_C5_WHOLE") == "1", reason="PJB_SKIP_C5_WHOLE=1")
def test_c5_whole_on_one_gpu():
    import torch
    from oracle import oracle as orc
    from portcullis_amd import ffi, synth

    import gc
    gc.collect()
    torch.cuda.empty_cache()  # (what earlier tests of this process left in torch's caching allocator)
    free, _ = torch.cuda.mem_get_info()
    print(f"free HBM at the start: {free / 1e9:.1f} GB")
    if free < 125e9:
        pytest.skip(f"needs ~120 GB of free HBM, {free / 1e9:.0f} GB are free")
    cfgs = synth.c3_contig_configs(1_000_000_000, 300_000)
    lens = [c.contig_len for c in cfgs]
    with ffi.Context(0, "FR", strandedness=1) as ctx:          # PJB_SS_FIRSTSTRAND: accepted, no effect on junc output
        ctx.set_refs(lens)
        data = []
        for tid, c in enumerate(cfgs):
            d = synth.generate(c, device="cuda", tid=tid)
            ctx.upload_contig_device(tid, d["genome"])
            data.append(d)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()  # the generator's temporaries go back to the driver: the library allocates with hipMalloc, not through torch
        n_reads = sum(d["n_reads"] for d in data)
        n_pairs = sum(d["n_pairs"] for d in data)
        print(f"records resident: torch allocated {torch.cuda.memory_allocated() / 1e9:.1f} GB, reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB, "
              f"free on the device {torch.cuda.mem_get_info()[0] / 1e9:.1f} GB")
        assert n_reads >= 999_000_000
        groups = ffi.plan_groups(lens, list(range(len(cfgs))))
        assert len(groups) == 3

        def run():
            # two chains queued at a time: a 335 M-read chain's scratch is ~37 GB, and this process also holds the other
            # full-size fixtures of the module (`bench.py --config c5` queues all three)
            ctx.clear_rows()
            regs, queued = {}, []
            for g in groups:
                for tid in g:
                    ctx.submit_batch_device(tid, data[tid]["batch"], data[tid]["n_reads"])
                ctx.finish_group_begin(g)
                queued.append(g)
                if len(queued) == 2:
                    regs.update(ctx.finish_group_end(queued.pop(0)))
            while queued:
                regs.update(ctx.finish_group_end(queued.pop(0)))
            return ctx.collect(), regs

        rows, regs = run()
        assert sum(r["n_reads"] for r in regs.values()) == n_reads and sum(r["n_pairs"] for r in regs.values()) == n_pairs
        assert sum(r["spliced"] + r["unspliced"] for r in regs.values()) == n_reads
        for tid, d in enumerate(data):
            assert regs[tid]["n_reads"] == d["n_reads"] and regs[tid]["n_pairs"] == d["n_pairs"]
            assert regs[tid]["n_junctions"] == int((rows["refid"] == tid).sum())
        assert int(rows["nb_raw"].astype(np.int64).sum()) == n_pairs            # conservation
        assert (rows["r1pos"] + rows["r1neg"] + rows["r2pos"] + rows["r2neg"] == rows["nb_raw"]).all()
        key = (rows["refid"].astype(np.int64) << 48) | (rows["start"].astype(np.int64) << 20)
        assert (np.diff(key) >= 0).all()                                          # target-major, start-sorted
        assert len(np.unique(rows[["refid", "start", "end"]])) == len(rows) and len(rows) > 290_000
        deep = rows[np.argmax(rows["nb_raw"])]
        assert int(deep["nb_raw"]) > 1_000_000
        again, _ = run()
        assert hashlib.md5(again.tobytes()).hexdigest() == hashlib.md5(rows.tobytes()).hexdigest()   # deterministic
        worst = 0.0
        for tid in sorted({0, int(deep["refid"]), 24}):
            hb = synth.batch_to_numpy(data[tid]["batch"], 0, data[tid]["n_reads"])
            orows, oreg = orc.find_juncs(tid, lens[tid], data[tid]["genome"].cpu().numpy().tobytes(), hb, "FR")
            region_equal(regs[tid], oreg)
            worst = max(worst, assert_rows_equal(rows[rows["refid"] == tid], orows))
            del hb
        assert worst <= 1e-6
