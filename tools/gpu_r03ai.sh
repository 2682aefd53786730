#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof -o d -- python3 $GRAFT_REPO_ROOT/tools/bench_deflate.py --workdir /tmp/pjb_dfl > $GRAFT_REPO_ROOT/gpurun_out/r03ai_deflate.json 2> /tmp/d.err
tail -2 /tmp/d.err
cat $GRAFT_REPO_ROOT/gpurun_out/r03ai_deflate.json
f=$(find /tmp/dprof -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-200
cp $f $GRAFT_REPO_ROOT/gpurun_out/r03ai_deflate_kernel_stats.csv
