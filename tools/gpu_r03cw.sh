#!/bin/bash
# the bench line and the bamfilt program at the last tree, timed runs starting on an idle box (kernels as in r03cg; the command's child process active under the harness)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py > gpurun_out/r03cx_bench_C3.json 2> gpurun_out/r03cx_bench.err
tail -c 400 gpurun_out/r03cx_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03cx_bench_C3.json'))
print('value',d['value'],'ms',d['ms_per_step'],'step_frac',d['roofline']['step_frac'])
print('e2e', d['e2e']['wall_s'], d['e2e']['runs_s'], 'cpu', d['e2e']['cpu']['wall_s'], d['e2e']['tab_identical_to_oracle'])
PY
python tools/bench_bamfilt_program.py --runs 9 > gpurun_out/r03cx_bamfilt_program.json 2>/dev/null; cut -c1-330 gpurun_out/r03cx_bamfilt_program.json
