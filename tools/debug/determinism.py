import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from portcullis_amd import ffi, synth, distributed as pd
dev = torch.device("cuda", 0)
cfgs = synth.c3_contig_configs(20_000_000, 25_000)
lens = [c.contig_len for c in cfgs]
shards = pd.shard_contigs([c.n_reads for c in cfgs], 2)
data = {t: synth.generate(cfgs[t], device=dev, tid=t) for t in range(25)}
def run(order, reps, flags=0):
    ctx = ffi.Context(0, "FR", flags=flags); ctx.set_refs(lens)
    for t in order: ctx.upload_contig_device(t, data[t]["genome"])
    out = {}
    for _ in range(reps):
        ctx.clear_rows()
        for t in order:
            ctx.submit_batch_device(t, data[t]["batch"], data[t]["n_reads"]); ctx.finish_contig(t)
        rows = ctx.collect()
        out = {t: rows[rows["refid"] == t].copy() for t in order}
    ctx.close(); return out
A = run(list(range(25)), 1)
for name, order, reps, fl in [("shard0 x3", shards[0], 3, 1), ("shard1 x3", shards[1], 3, 1), ("all x2", list(range(25)), 2, 0), ("reverse", list(range(24, -1, -1)), 1, 0)]:
    B = run(order, reps, fl)
    for t in order:
        a, b = A[t], B[t]
        if a.tobytes() != b.tobytes():
            print(name, "contig", t, "differs: len", len(a), len(b))
            if len(a) == len(b):
                for f in a.dtype.names:
                    bad = np.nonzero((a[f] != b[f]).reshape(len(a), -1).any(1))[0]
                    if bad.size: print("   field", f, bad.size, "rows; first", bad[0], a[f][bad[0]], b[f][bad[0]], "key", a["start"][bad[0]], a["end"][bad[0]], "raw", a["nb_raw"][bad[0]])
    print(name, "done")
