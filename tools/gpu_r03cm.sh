#!/bin/bash
# chains queued at once: 3 (the default) against 2, alternating on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in 1 2 3; do
for q in 3 2; do
timeout 600 python bench.py --no-cpu-baseline --no-e2e --queue $q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('queue $q:', round(d['ms_per_step'],3), 'ms; kernels', d['device_kernel_ms_per_step'])"
done
done | tee gpurun_out/r03cm_queue.txt
