#!/bin/bash
# runtime knobs and the configs[2] step: kernel arguments in device memory
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1 0 1; do
export HIP_FORCE_DEV_KERNARG=$v
timeout 600 python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('HIP_FORCE_DEV_KERNARG=$v:', round(d['ms_per_step'],3), 'ms; host queueing', d['host_queue_ms_per_step'], 'kernels', d['device_kernel_ms_per_step'])"
done | tee gpurun_out/r03cs_kernarg.txt
