"""One configs[2]-sized contig (8 M paired-end reads), pjb_finish_contig in a loop: per-kernel durations of the first
kernels of the chain with nothing beside them.  PJB_FUSED_K1 / PJB_K1W_DEBUG select the variant."""
import sys, os, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from portcullis_amd import ffi, synth
n = int(os.environ.get("K1_READS", 8_000_000))
cfg = dataclasses.replace(synth.CONFIGS["C2"], n_reads=n, contig_len=15 * n, n_junctions=n // 800, read_len=150, paired=True)
d = synth.generate(cfg, device="cuda")
torch.cuda.synchronize()
with ffi.Context(0, "FR", flags=ffi.FLAG_KERNEL_TIMING) as ctx:
    ctx.set_refs([cfg.contig_len]); ctx.upload_contig_device(0, d["genome"])
    ctx.set_option("overlap", 0)
    for _ in range(2):
        ctx.clear_rows(); ctx.submit_batch_device(0, d["batch"], n)
        try: ctx.finish_contig(0)
        except Exception as e: print("err", e)
    ctx.reset_kernel_timing()
    R = 5
    for _ in range(R):
        ctx.clear_rows(); ctx.submit_batch_device(0, d["batch"], n)
        try: ctx.finish_contig(0)
        except Exception as e: pass
    kt = ctx.kernel_timing()
    line = " ".join(f"{k}={v[1]/v[0]*1e3:.0f}us" for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1]) if k.startswith("k1") and v[0])
    print(f"fused={os.environ.get('PJB_FUSED_K1','1')} dbg={os.environ.get('PJB_K1W_DEBUG','0')} N={n/1e6:.0f}M P={d['n_pairs']/1e6:.2f}M S={d['n_spliced']/1e6:.2f}M | {line}", flush=True)
