#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_groups.py tests/test_gpu_extra.py tests/test_gpu_ingest.py -x -q 2>&1 | tail -4 )
for k in 1 2; do
python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 > gpurun_out/r03an_b.json 2>> gpurun_out/r03an_bench.err
python - <<'PY' | tee -a gpurun_out/r03an.txt
import json
d = json.loads(open('gpurun_out/r03an_b.json').read().strip().split('\n')[-1])
r = d['roofline']
print('ms/step', round(d['ms_per_step'], 2), 'step_frac', r.get('step_frac'), r['kernel'], r.get('avg_kernel_ms'), r.get('avg_kernel_ms_alone'), r.get('frac_alone'))
print('  ', [(k['name'], k['avg_ms'], round(k['ms_per_step'],2)) for k in d['kernels'][:14]])
PY
done
python tools/bench_extra.py 2>/dev/null | cut -c1-420
