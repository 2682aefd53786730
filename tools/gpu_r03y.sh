#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 > gpurun_out/r03y_b.json 2>> gpurun_out/r03y_bench.err
  python - "$label" <<'PY' | tee -a gpurun_out/r03y_streams.txt
import json, sys
d = json.loads(open('gpurun_out/r03y_b.json').read().strip().split('\n')[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'], 2), 'step_frac', d['roofline'].get('step_frac'))
PY
}
run "main first, 4 eager" A=1
run "side first, 4 eager" PJB_SLOT_SIDE_FIRST=1
run "main first, 8 eager" PJB_EAGER_SLOTS=8
run "main first, 3 eager" PJB_EAGER_SLOTS=3
run "main first, 4 eager again" A=1
run "side first, 4 eager again" PJB_SLOT_SIDE_FIRST=1
