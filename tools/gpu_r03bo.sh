#!/bin/bash
# k1_scan_tiles over a few blocks: parity suites that go through it + the configs[2] step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groups.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
python tests/fuzz_campaign.py --seeds 60 2>&1 | tail -2
python tests/fuzz_groups.py --seeds 30 2>&1 | tail -2
timeout 900 python bench.py --no-cpu-baseline --no-e2e > gpurun_out/r03bo_bench.json 2> gpurun_out/r03bo_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03bo_bench.json').read().strip().split('\n')[-1])
print('ms', d['ms_per_step'], 'frac', d['roofline']['step_frac'])
for k in d['kernels'][:12]: print(k['name'], k['avg_ms'], k['ms_per_step'])
PY
