#!/bin/bash
# round 3, call c: bgzf_decode with asynchronous ring refill + one-store tokens, bgzf_resolve with exact dependency ranges
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_ingest.py -q -x 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r03c_pytest.log
run() { # tag, times
  ( timeout 600 python tools/bench_inflate.py --times $2 --chunk-mb 16384 > gpurun_out/r03c_inflate_$1.json 2> gpurun_out/r03c_inflate_$1.err ) 2>&1 | tail -3
  tail -c 300 gpurun_out/r03c_inflate_$1.err; python - <<PY
import json
d=json.load(open('gpurun_out/r03c_inflate_$1.json'))
print('$1', d['blocks'], 'blocks', d['kernel_ms'], 'ms', d['kernel_gbps_inflated'], 'GB/s', d['kernels'])
PY
}
run t2 2
run t3 3
run t5 5
for lits in 2 4; do
  make -B -C portcullis_amd/csrc HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DPJB_I2_LITS=$lits" > /dev/null 2>&1
  run lits${lits}_t2 2
done
make -B -C portcullis_amd/csrc > /dev/null 2>&1
