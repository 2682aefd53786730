#!/bin/bash
# quick GPU round for the one-pass K1: parity tests that exercise it, then the bench without the CPU / e2e legs, both ways
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_extra.py -m gpu -q -x 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/k1_pytest.log
for f in 1 0; do
  PJB_FUSED_K1=$f timeout 600 python bench.py --no-e2e --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/k1_bench_$f.json 2> gpurun_out/k1_bench_$f.err
  tail -c 300 gpurun_out/k1_bench_$f.err
  python - <<PY
import json
d=json.load(open("gpurun_out/k1_bench_$f.json"))
print("fused=$f", round(d["ms_per_step"],3), "ms/step", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_alone"), d["roofline"]["step_frac"])
for k in d["kernels"][:8]:
    print(f'  {k["name"]:20s} {k["avg_ms"]*1e3:8.1f}us {k["ms_per_step"]:7.3f} {k["gbps"]}')
PY
done
