#!/bin/bash
# --extra per-kernel breakdown; queue depth sweep of the bench step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/debug/extra_breakdown.py > gpurun_out/r03x_extra_breakdown.txt 2>&1
for q in 3 4 5 7; do
  python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 --queue $q > gpurun_out/r03x_bench_q$q.json 2>> gpurun_out/r03x_bench.err
  python - $q <<'PY' | tee -a gpurun_out/r03x_queue.txt
import json, sys
d = json.loads(open(f'gpurun_out/r03x_bench_q{sys.argv[1]}.json').read().strip().split('\n')[-1])
print('queue', sys.argv[1], 'ms/step', round(d['ms_per_step'], 2), 'step_frac', d['roofline'].get('step_frac'))
PY
done
