#!/bin/bash
# usage: tools/profile_round.sh <tag>   (run under gpurun): kernel-trace stats + PMC traffic of the bench command
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT && bash tools/pmc_traffic.sh $TAG
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
tail -c 300 $OUT/bench_$TAG.err
