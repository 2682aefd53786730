#!/bin/bash
# runtime knobs and the configs[2] step: completion signals polled instead of interrupt-driven
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 0 1 0; do
export HSA_ENABLE_INTERRUPT=$v
timeout 600 python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('HSA_ENABLE_INTERRUPT=$v:', round(d['ms_per_step'],3), 'ms; kernels', d['device_kernel_ms_per_step'])"
done | tee gpurun_out/r03ct_interrupt.txt
