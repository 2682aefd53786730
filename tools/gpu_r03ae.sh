#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_bamfilt.py -x -q 2>&1 | tail -5
python tools/bench_bamfilt_program.py --runs 3 > gpurun_out/r03ae_bamfilt_program.json 2> gpurun_out/r03ae_bamfilt_program.err; tail -3 gpurun_out/r03ae_bamfilt_program.err; cat gpurun_out/r03ae_bamfilt_program.json
python tools/bench_deflate.py > gpurun_out/r03ae_deflate.json 2> gpurun_out/r03ae_deflate.err; tail -3 gpurun_out/r03ae_deflate.err; cat gpurun_out/r03ae_deflate.json
