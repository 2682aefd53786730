for a in 0 1 2 3; do echo "ablate $a"; PJB_ABLATE=$a python - <<'PY'
import sys, os
sys.path.insert(0, '.')
import torch
from portcullis_amd import ffi, synth
cfg = synth.CONFIGS["C2"]
d = synth.generate(cfg, "cuda")
ctx = ffi.Context(0, flags=1)
ctx.set_refs([cfg.contig_len]); ctx.upload_contig_device(0, d["genome"])
for i in range(6):
    if i == 2: ctx.reset_kernel_timing()
    ctx.clear_rows(); ctx.submit_batch_device(0, d["batch"], d["n_reads"])
    try: ctx.finish_contig(0)
    except Exception as e: pass
kt = ctx.kernel_timing()
print({k: round(v[1]/max(v[0],1),4) for k,v in kt.items() if k in ("k4_pairs","k1_emit","k5_finalize","k1_count")})
PY
done
