import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from portcullis_amd import ffi, synth
import bench_extra as be
data = be.make(synth.CONFIGS["C2"], synth, torch)
lens = [synth.CONFIGS["C2"].contig_len] * 2
for flags, label in ((0, "plain"), (ffi.FLAG_EXTRA, "extra")):
    with ffi.Context(0, "UNKNOWN", flags=flags) as ctx:
        ctx.set_refs(lens)
        for tid, d in enumerate(data): ctx.upload_contig_device(tid, d["genome"])
        for rep in range(4):
            ctx.clear_rows(); torch.cuda.synchronize()
            t0 = time.perf_counter(); marks = []
            def m(what): marks.append((what, round((time.perf_counter() - t0) * 1e3, 3)))
            for tid, d in enumerate(data):
                ctx.submit_batch_device(tid, d["batch"], d["n_reads"]); m(f"submit{tid}")
                ctx.finish_contig_begin(tid); m(f"begin{tid}")
            for tid in range(2):
                ctx.finish_contig_end(tid); m(f"end{tid}")
            ctx.collect(copy=False); m("collect")
            if flags: ctx.extra_finish(copy=False); m("finish")
        print(label, marks)
