#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 > gpurun_out/r03z_b.json 2>> gpurun_out/r03z_bench.err
  python - "$label" <<'PY' | tee -a gpurun_out/r03z_streams.txt
import json, sys
d = json.loads(open('gpurun_out/r03z_b.json').read().strip().split('\n')[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'], 2), 'step_frac', d['roofline'].get('step_frac'))
PY
}
run "default (m0 s0 m1 s1 ...)" A=1
run "x + default" PJB_STREAM_PLAN=x
run "xx + default" PJB_STREAM_PLAN=xx
run "xxx + default" PJB_STREAM_PLAN=xxx
run "mains first" PJB_STREAM_PLAN=m0m1m2m3s0s1s2s3
run "x mains first" PJB_STREAM_PLAN=xm0m1m2m3s0s1s2s3
run "xx mains first" PJB_STREAM_PLAN=xxm0m1m2m3s0s1s2s3
run "xxx mains first" PJB_STREAM_PLAN=xxxm0m1m2m3s0s1s2s3
run "mains first, sides rotated" PJB_STREAM_PLAN=m0m1m2m3s2s3s0s1
run "x mains first, sides rotated" PJB_STREAM_PLAN=xm0m1m2m3s2s3s0s1
run "mains first, sides rotated 1" PJB_STREAM_PLAN=m0m1m2m3s1s2s3s0
run "hwq 8, mains first" GPU_MAX_HW_QUEUES=8 PJB_STREAM_PLAN=m0m1m2m3s0s1s2s3
run "hwq 8, x mains first" GPU_MAX_HW_QUEUES=8 PJB_STREAM_PLAN=xm0m1m2m3s0s1s2s3
run "hwq 2" GPU_MAX_HW_QUEUES=2
