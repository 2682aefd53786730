#!/bin/bash
# round 3, call m: whole GPU suite (incl. groups, two device threads on one GPU, configs[4] whole), then the step with 25 / 7 / 3 chains
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r03m_pytest.log
for gb in 0 536870912 1073741824; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --group-bases $gb > gpurun_out/r03m_bench_${gb}.json 2> gpurun_out/r03m_bench.err || tail -c 300 gpurun_out/r03m_bench.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03m_bench_${gb}.json'))
print('group-bases $gb chains', len(d['config']['chains']), 'ms/step %.2f' % d['ms_per_step'], 'kernel ms/step %.2f' % d['device_kernel_ms_per_step'], 'step_frac', d['roofline']['step_frac'])
for k in d['kernels'][:12]: print('   %-18s %4.0f x %8.1f us = %6.3f ms' % (k['name'], k['launches_per_step'], k['avg_ms'] * 1000, k['ms_per_step']), k['gbps'])
PY
done 2>&1 | tee gpurun_out/r03m_sweep.txt
