#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extra.py -x -q 2>&1 | tail -25 > gpurun_out/r03aa_pytest.log
cat gpurun_out/r03aa_pytest.log
python tools/bench_extra.py > gpurun_out/r03aa_extra.json 2> gpurun_out/r03aa_extra.err; tail -3 gpurun_out/r03aa_extra.err; cut -c1-700 gpurun_out/r03aa_extra.json
python tools/debug/extra_breakdown.py > gpurun_out/r03aa_extra_breakdown.txt 2>&1; tail -32 gpurun_out/r03aa_extra_breakdown.txt
