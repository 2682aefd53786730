#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extra.py -x -q 2>&1 | tail -5
python tools/debug/extra_marks.py 2>&1 | grep -v amdgpu.ids | tail -28 | tee gpurun_out/r03ac.txt
python tools/bench_extra.py 2>/dev/null | cut -c1-600
