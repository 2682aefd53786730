#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15
( time python bench.py --steps 20 --warmup 3 > gpurun_out/r02b_bench_C3.json 2> gpurun_out/r02b_bench.err ) 2>&1 | tail -4
tail -c 400 gpurun_out/r02b_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02b_bench_C3.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["e2e"])
PY
