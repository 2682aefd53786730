#!/bin/bash
# configs[2] step: queue depth and group size once more, now that k1_scan_tiles is no longer a chain's longest serial kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for gb in 268435456 536870912 805306368 1073741824; do
for q in 3 4; do
timeout 600 python bench.py --no-cpu-baseline --no-e2e --queue $q --group-bases $gb 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('group-bases $gb queue $q:', round(d['ms_per_step'],3), 'ms', d['config']['chains'])"
done
done | tee gpurun_out/r03by_queue_groups.txt
