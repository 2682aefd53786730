#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_groups.py tests/test_gpu_ingest.py -x -q 2>&1 | tail -6 )
( timeout 900 python tests/fuzz_campaign.py --seeds 200 --start 61000 2>&1 | tail -3 )
python tools/bench_long_reads.py 2>/dev/null | cut -c1-700
python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 > gpurun_out/r03bg_b.json 2>> gpurun_out/r03bg_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03bg_b.json').read().strip().split('\n')[-1])
print('ms/step', round(d['ms_per_step'], 2), 'step_frac', d['roofline'].get('step_frac'))
print('  ', [(k['name'], k['avg_ms'], round(k['ms_per_step'],2)) for k in d['kernels'][:6]])
PY
