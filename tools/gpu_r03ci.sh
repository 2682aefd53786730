#!/bin/bash
# how much of a configs[2] step does the calling thread spend queueing its ~800 launches?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for q in 3 2; do
timeout 600 python bench.py --no-cpu-baseline --no-e2e --queue $q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('queue $q:', round(d['ms_per_step'],3), 'ms; host queueing', d['host_queue_ms_per_step'], 'ms; kernels', d['device_kernel_ms_per_step'], 'overlap', d['overlap_factor'])"
done
