#!/bin/bash
# round 3, call b: the two-kernel inflate (bgzf_decode + bgzf_resolve) against round 2's bgzf_inflate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_ingest.py -q -x 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r03b_pytest.log
for v in 0 1; do
  ( time PJB_INFLATE_V1=$v timeout 600 python tools/bench_inflate.py --times 2 --chunk-mb 8192 > gpurun_out/r03b_inflate_v1_$v.json 2> gpurun_out/r03b_inflate_v1_$v.err ) 2>&1 | tail -3
  tail -c 300 gpurun_out/r03b_inflate_v1_$v.err; cat gpurun_out/r03b_inflate_v1_$v.json | cut -c1-700
done
