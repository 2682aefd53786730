"""Child process of tests/test_gpu_poison.py (run with PJB_POISON=1): chains of very different sizes through three control slots."""
import sys
import os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import torch  # (before the library: see conftest.py)
from fuzzgen import make_reads, to_batch
from oracle import oracle as orc
from parity import assert_rows_equal, region_equal
from portcullis_amd import ffi
# slots are reused by smaller and by larger chains, with several chains in flight: what a chain reads behind its own data
# is what an earlier one left there -- or the pattern
sizes = [5000, 900, 7000, 300, 64, 1, 6500, 2500, 8000, 120]
contigs = []
for k, n in enumerate(sizes):
    genome, reads = make_reads(700 + k, n_reads=n, paired=True, glen=20000 + 3000 * k)
    batch = to_batch(reads)
    rows, reg = orc.find_juncs(k, len(genome), genome, batch, "FR")
    contigs.append((genome, batch, rows, reg))
with ffi.Context(0, "FR") as ctx:
    ctx.set_refs([len(c[0]) for c in contigs])
    for rounds in range(2):
        ctx.clear_rows()
        queued = []
        got = {}
        def collect():
            t = queued.pop(0)
            got[t] = ctx.finish_contig_end(t)
        for t, (genome, batch, rows, reg) in enumerate(contigs):
            ctx.upload_contig(t, genome.encode())
            ctx.submit_batch(t, batch)
            ctx.finish_contig_begin(t)
            queued.append(t)
            if len(queued) >= 3:
                collect()
        while queued:
            collect()
        import numpy as np
        all_rows = ctx.collect()
        want = np.concatenate([c[2] for c in contigs])
        assert_rows_equal(all_rows, want)
        for t, c in enumerate(contigs):
            region_equal(got[t], c[3])
print("poison ok")
