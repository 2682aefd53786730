#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
for q in 3 4 5 6 8; do
  timeout 600 python bench.py --no-e2e --no-cpu-baseline --steps 10 --warmup 2 --queue $q > gpurun_out/q_bench_$q.json 2> gpurun_out/q_bench_$q.err
  python - <<PY
import json
d=json.load(open("gpurun_out/q_bench_$q.json"))
print("queue=$q", round(d["ms_per_step"],3), "ms/step", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_frac"])
PY
done
