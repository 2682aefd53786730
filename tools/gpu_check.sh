#!/bin/bash
# Full GPU test suite + the default bench line, as the driver runs them.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/check_pytest.log
( time python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/check_smoke.log
( time python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err ) 2>&1 | tail -4
tail -c 600 gpurun_out/check_bench.err
cat gpurun_out/check_bench.json
