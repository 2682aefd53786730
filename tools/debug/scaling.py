import sys, os, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from portcullis_amd import ffi, synth
paired = len(sys.argv) > 1 and sys.argv[1] == "pe"
for n in ((10_000_000,) if len(sys.argv) > 2 else (2_000_000, 8_000_000, 16_000_000, 32_000_000, 64_000_000)):
    cfg = dataclasses.replace(synth.CONFIGS["C2"], n_reads=n, contig_len=10 * n, n_junctions=n // (800 if paired else 200), read_len=150 if paired else 100, paired=paired)
    d = synth.generate(cfg, device="cuda")
    torch.cuda.synchronize()
    with ffi.Context(0, "FR" if paired else "UNKNOWN", flags=ffi.FLAG_KERNEL_TIMING) as ctx:
        ctx.set_refs([cfg.contig_len]); ctx.upload_contig_device(0, d["genome"])
        for _ in range(2):
            ctx.clear_rows(); ctx.submit_batch_device(0, d["batch"], n); ctx.finish_contig(0)
        ctx.reset_kernel_timing()
        R = 5
        import time
        t0 = time.perf_counter()
        for _ in range(R):
            ctx.clear_rows(); ctx.submit_batch_device(0, d["batch"], n); reg = ctx.finish_contig(0)
        dt = (time.perf_counter() - t0) / R
        kt = ctx.kernel_timing()
        P = d["n_pairs"]
        line = " ".join(f"{k}={v[1]/v[0]*1e3:.0f}us(x{v[0]//R})" for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1])[:9])
        print(f"N={n/1e6:.0f}M P={P/1e6:.2f}M J={reg['n_junctions']} step={dt*1e3:.2f}ms ns/pair={dt/P*1e9:.2f} passes={ctx.timing()['sort_passes']} | {line}", flush=True)
    del d
