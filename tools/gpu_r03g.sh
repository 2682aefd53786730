#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 tools/debug/dma_probe 2>&1 | tee gpurun_out/r03g_dma_probe.txt
( time timeout 900 python -m pytest tests/test_gpu_ingest.py -q -x 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r03g_pytest.log
run() { # tag, times
  ( timeout 600 python tools/bench_inflate.py --times $2 --chunk-mb 16384 > gpurun_out/r03g_inflate_$1.json 2> gpurun_out/r03g_inflate_$1.err ) 2>&1 | tail -3
  tail -c 300 gpurun_out/r03g_inflate_$1.err; python - <<PY
import json
d=json.load(open('gpurun_out/r03g_inflate_$1.json'))
print('$1', d['blocks'], 'blocks', d['kernel_ms'], 'ms', d['kernel_gbps_inflated'], 'GB/s', d['kernels'])
PY
}
run t2 2
run t5 5
