#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 $EXTRA > gpurun_out/r03aj_b.json 2>> gpurun_out/r03aj_bench.err
  python - "$label" <<'PY' | tee -a gpurun_out/r03aj_queue.txt
import json, sys
d = json.loads(open('gpurun_out/r03aj_b.json').read().strip().split('\n')[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'], 2), 'step_frac', d['roofline'].get('step_frac'))
PY
}
EXTRA="--queue 3" run "queue 3" A=1
EXTRA="--queue 4" run "queue 4" A=1
EXTRA="--queue 5" run "queue 5" A=1
EXTRA="--queue 4" run "queue 4, hwq 8" GPU_MAX_HW_QUEUES=8
EXTRA="--queue 5" run "queue 5, hwq 8" GPU_MAX_HW_QUEUES=8
EXTRA="--queue 2" run "queue 2" A=1
EXTRA="--queue 4 --group-bases 268435456" run "queue 4, 0.25 Gb groups" A=1
EXTRA="--queue 3 --group-bases 268435456" run "queue 3, 0.25 Gb groups" A=1
