#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
PJB_BENCH_KEEP_WORKDIR=1 PJB_BENCH_E2E_REPS=7 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r03as_bench.json 2> gpurun_out/r03as_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03as_bench.json').read().strip().split('\n')[-1])
print('ms', d['ms_per_step'], 'e2e', d['e2e']['wall_s'], d['e2e']['runs_s'])
PY
W=/tmp/pjb_bench_e2e
rm -rf /tmp/e2e_prof
export PJB_NORMAL_EXIT=1
( time rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/e2e_prof -- portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc3 $W/prep > /dev/null 2> gpurun_out/r03as_rocprof.err ) 2>&1 | tail -3
python tools/debug/e2e_timeline.py /tmp/e2e_prof > gpurun_out/r03as_e2e_timeline.txt 2>&1
head -40 gpurun_out/r03as_e2e_timeline.txt
