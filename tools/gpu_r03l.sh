#!/bin/bash
# round 3, call l: group tests again, then chain size / queue depth sweep of the bench step (no e2e), then the full line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_groups.py -q -x 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r03l_pytest.log
for gb in 0 268435456 536870912 1073741824; do for q in 3 4; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --group-bases $gb --queue $q > gpurun_out/r03l_bench_${gb}_$q.json 2> gpurun_out/r03l_bench.err || tail -c 300 gpurun_out/r03l_bench.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03l_bench_${gb}_$q.json'))
print('group-bases $gb queue $q chains', len(d['config']['chains']), 'ms/step %.2f' % d['ms_per_step'], 'kernel ms/step %.2f' % d['device_kernel_ms_per_step'], 'step_frac', d['roofline']['step_frac'])
PY
done; done 2>&1 | tee gpurun_out/r03l_sweep.txt
