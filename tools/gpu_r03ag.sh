#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_deflate.py -x -q -s 2>&1 | tail -4
python tools/bench_deflate.py --workdir /tmp/pjb_dfl > gpurun_out/r03ag_deflate.json 2> gpurun_out/r03ag_deflate.err; tail -3 gpurun_out/r03ag_deflate.err; cat gpurun_out/r03ag_deflate.json
