#!/usr/bin/env python3
"""What every rank's share of BASELINE configs[2] costs on ONE GPU (bench.py with PJB_BENCH_AS_RANK=r/N, no exchange): the step of rank r
of N for N = 1, 2, 3, 4, 8 and EVERY r -- an N-GPU step is the slowest rank's, so  N=1 step / (N * max_r step)  bounds the strong-scaling
efficiency from above.  Never a measurement of N GPUs.  Writes gpurun_out/<TAG>_rank_share.json (copied to profiles/ by hand).
    python tools/rank_share.py [TAG] [N,N,...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
ns = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4, 8]
sys.path.insert(0, os.path.join(ROOT, "tools"))
from csrc_hash import csrc_hash  # noqa: E402

out = {"_what": __doc__.strip().split("\n    python")[0], "_csrc_hash": csrc_hash(ROOT), "_steps": 10, "_warmup": 3, "shares": {}}
for n in ns:
    rows = []
    for r in range(n):
        env = dict(os.environ)
        if n > 1:
            env["PJB_BENCH_AS_RANK"] = f"{r}/{n}"
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-e2e", "--no-cpu-baseline",
                            "--no-back-to-back"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception:
            rows.append({"rank": r, "error": p.stderr[-400:]})
            continue
        rows.append({"rank": r, "ms_per_step": round(d["ms_per_step"], 3), "reads": d["config"]["reads_rank0"], "chains": d["config"]["chains"],
                     "targets": d["config"].get("targets_rank"), "launches_per_step": d["launches_per_step"], "overlap_factor": d["overlap_factor"],
                     "device_kernel_ms_per_step": d["device_kernel_ms_per_step"],
                     "chain_stages_ms_alone": d.get("chain_stages_ms_alone")})
        print(n, rows[-1], flush=True)
    out["shares"][str(n)] = rows
whole = out["shares"].get("1", [{}])[0].get("ms_per_step")
out["bound"] = {}
for n, rows in out["shares"].items():
    ms = [x["ms_per_step"] for x in rows if "ms_per_step" in x]
    if whole and ms and len(ms) == int(n):
        out["bound"][n] = {"worst_rank_ms": max(ms), "best_rank_ms": min(ms), "strong_scaling_efficiency_upper_bound": round(whole / (int(n) * max(ms)), 3)}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_rank_share.json"), "w"), indent=1)
print(json.dumps(out["bound"], indent=1))
