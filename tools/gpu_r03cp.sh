#!/bin/bash
# the configs[2] step over the runtime's number of hardware queues (the program sets 8 for itself; the bench's process had the default, 4)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 3 5 6 default 3 5; do
if [ $v = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$v; fi
timeout 600 python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('GPU_MAX_HW_QUEUES $v:', round(d['ms_per_step'],3), 'ms; kernels', d['device_kernel_ms_per_step'])"
done | tee gpurun_out/r03cp_hw_queues2.txt
