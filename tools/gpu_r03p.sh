#!/bin/bash
# round 3, call p: dense k1_emit: parity suites, configs[4] whole, the step, the full line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_groups.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_extra.py -q -x 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r03p_pytest.log
( time timeout 1800 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r03p_pytest_full.log
for gb in 0 536870912; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --group-bases $gb > gpurun_out/r03p_bench_${gb}.json 2> gpurun_out/r03p_bench.err || tail -c 300 gpurun_out/r03p_bench.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03p_bench_${gb}.json'))
print('group-bases $gb chains', len(d['config']['chains']), 'ms/step %.2f' % d['ms_per_step'], 'kernel ms/step %.2f' % d['device_kernel_ms_per_step'], 'step_frac', d['roofline']['step_frac'], d['roofline']['kernel'], d['roofline'].get('frac'), d['roofline'].get('frac_alone'))
for k in d['kernels'][:8]: print('   %-18s %4.0f x %8.1f us = %6.3f ms' % (k['name'], k['launches_per_step'], k['avg_ms'] * 1000, k['ms_per_step']), k['gbps'])
PY
done 2>&1 | tee gpurun_out/r03p_sweep.txt
( time timeout 1500 python bench.py --config c5 --steps 5 --warmup 2 > gpurun_out/r03p_bench_c5.json 2> gpurun_out/r03p_bench_c5.err ) 2>&1 | tail -3
tail -c 400 gpurun_out/r03p_bench_c5.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r03p_bench_c5.json'))
    print('c5 value',d['value'],'ms',d['ms_per_step'],'step_frac',d['roofline']['step_frac'], d['config']['workload'][:160], 'cpu', json.dumps(d['cpu_baseline'])[:400])
except Exception as e: print('c5 bench failed', e)
PY
