#!/bin/bash
# round 3, call a: GPU tests, the bench line (e2e median of 5 + the CPU neighbour), the inflate bench -- the baseline of the round
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r03a_pytest.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err ) 2>&1 | tail -4
tail -c 400 gpurun_out/r03a_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03a_bench.json'))
print('value',d['value'],'ms',d['ms_per_step'],'step_frac',d['roofline']['step_frac'])
print('e2e',json.dumps(d['e2e'])[:1800])
PY
( time timeout 600 python tools/bench_inflate.py --times 1 > gpurun_out/r03a_inflate.json 2> gpurun_out/r03a_inflate.err ) 2>&1 | tail -3
cat gpurun_out/r03a_inflate.json
