import sys, os, time, dataclasses
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, numpy as np
from portcullis_amd import ffi, synth
import bench_extra as be
data = be.make(synth.CONFIGS["C2"], synth, torch)
lens = [synth.CONFIGS["C2"].contig_len] * 2
with ffi.Context(0, "UNKNOWN", flags=ffi.FLAG_EXTRA | ffi.FLAG_KERNEL_TIMING) as ctx:
    ctx.set_refs(lens)
    for tid, d in enumerate(data): ctx.upload_contig_device(tid, d["genome"])
    for rep in range(3):
        ctx.clear_rows(); ctx.reset_kernel_timing(); torch.cuda.synchronize()
        t0 = time.perf_counter(); marks = []
        for tid, d in enumerate(data):
            ctx.submit_batch_device(tid, d["batch"], d["n_reads"]); ctx.finish_contig_begin(tid)
        for tid, d in enumerate(data):
            ctx.finish_contig_end(tid); marks.append(time.perf_counter() - t0)
        ctx.collect(copy=False); xr = ctx.extra_finish(); marks.append(time.perf_counter() - t0)
    print("marks ms", [round(m * 1e3, 2) for m in marks])
    kt = ctx.kernel_timing()
    tot = 0
    for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{k:24s} {v[0]:4d} {v[1]:8.3f} ms"); tot += v[1]
    print("kernel total", sum(v[1] for v in kt.values()))
