#!/bin/bash
# run under gpurun: kernel trace of a short bench run, then the timeline of its last step
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --no-back-to-back > $OUT/tl_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/debug/timeline.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) ${1:-3} | tee $OUT/timeline.txt
