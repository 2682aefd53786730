#!/bin/bash
# does setting the kernel attributes beside the stream creation move the configs[2] step?  (hardware queues are handed out in creation order)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in 1 2; do
for v in serial beside; do
if [ $v = serial ]; then export PJB_CREATE_SERIAL=1; else unset PJB_CREATE_SERIAL; fi
timeout 600 python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v:', round(d['ms_per_step'],3), 'ms')"
done
done | tee gpurun_out/r03ch_create_order.txt
