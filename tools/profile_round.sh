#!/bin/bash
# usage: tools/profile_round.sh <tag> [commit]   (run under gpurun): rocprofv3 kernel-trace stats + PMC traffic of the
# bench command (BASELINE configs[2]), then the plain bench line.  Summaries land in gpurun_out/; copy to profiles/.
TAG=$1
COMMIT=${2:-unknown}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-back-to-back > $OUT/bench_prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_rocprof.py $(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_C3_kernel_stats.csv $OUT/${TAG}_rocprof.json C3 $COMMIT
bash tools/pmc_traffic.sh $TAG
python3 tools/summarize_pmc.py $OUT/pmc_$TAG $OUT/${TAG}_pmc_traffic.json C3 $COMMIT keep_last=2/7
python3 bench.py --steps 20 --warmup 3 ${BENCH_FLAGS:-} > $OUT/${TAG}_bench_C3.json 2> $OUT/bench_$TAG.err
tail -c 300 $OUT/bench_$TAG.err
# keep what travels back small: the raw traces stay on the box
rm -rf $OUT/prof_$TAG/*/*.db $OUT/prof_$TAG/*/*_kernel_trace.csv $OUT/pmc_$TAG/*/*/pmc_kernel_trace.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find $OUT -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
du -sh $OUT | tail -1
