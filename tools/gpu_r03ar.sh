#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_cli.py tests/test_gpu_bamfilt.py -x -q 2>&1 | tail -3
python tools/bench_bamfilt_program.py --runs 5 | tee gpurun_out/r03ar_bamfilt_program.json | cut -c1-330
PJB_BENCH_E2E_REPS=7 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r03ar_bench.json 2> gpurun_out/r03ar_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03ar_bench.json').read().strip().split('\n')[-1])
print('ms', d['ms_per_step'], 'e2e', d['e2e']['wall_s'], d['e2e']['runs_s'])
PY
