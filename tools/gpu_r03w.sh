#!/bin/bash
# e2e at HEAD (7 CLI runs), bamfilt program timing, --extra kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_bamfilt_program.py --runs 3 > gpurun_out/r03w_bamfilt_program.json 2> gpurun_out/r03w_bamfilt_program.err
python tools/bench_extra.py > gpurun_out/r03w_extra.json 2> gpurun_out/r03w_extra.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/xprof -o x -- python3 $GRAFT_REPO_ROOT/tools/bench_extra.py > /dev/null 2>&1 )
cp $(find /tmp/xprof -name "*kernel_stats.csv" | head -1) gpurun_out/r03w_extra_kernel_stats.csv
PJB_BENCH_E2E_REPS=7 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r03w_bench.json 2> gpurun_out/r03w_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03w_bench.json').read().strip().split('\n')[-1])
print('ms', d['ms_per_step'], 'e2e', d['e2e']['wall_s'], d['e2e']['runs_s'])
PY
