#!/bin/bash
# HBM traffic counters for the pjb kernels: separate --pmc passes (FETCH_SIZE takes 3 TCC slots,
# WRITE_SIZE 2), kernel trace only.  Results are summarised by tools/summarize_pmc.py, which averages the dispatches of the two
# timed steps only (keep_last=2/7: three warm-up passes and bench.py's two instrumented passes come first).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$ctr -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-e2e --no-back-to-back > $OUT/bench_$ctr.log 2>&1 || true
  ls $OUT/$ctr | head
done
