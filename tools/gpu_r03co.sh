#!/bin/bash
# experiment: every chain slot's streams on a part of the chip of their own (CU masks)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in none xcd cu none xcd cu; do
if [ $v = none ]; then unset PJB_CU_PARTITION; else export PJB_CU_PARTITION=$v; fi
timeout 600 python bench.py --no-cpu-baseline --no-e2e 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v:', round(d['ms_per_step'],3), 'ms; kernels', d['device_kernel_ms_per_step'], [(k['name'], k['avg_ms']) for k in d['kernels'][:4]])
except Exception as e: print('$v: failed', e)"
done | tee gpurun_out/r03co_cu_partition.txt
