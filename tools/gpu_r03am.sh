#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 $EXTRA > gpurun_out/r03am_b.json 2>> gpurun_out/r03am_bench.err
  python - "$label" <<'PY' | tee -a gpurun_out/r03am.txt
import json, sys
d = json.loads(open('gpurun_out/r03am_b.json').read().strip().split('\n')[-1])
ks = {k['name']: k for k in d['kernels']}
print(sys.argv[1], 'ms/step', round(d['ms_per_step'], 2), 'step_frac', d['roofline'].get('step_frac'), 'k1_emit', ks['k1_emit']['avg_ms'], 'k4b', ks['k4b_generic']['avg_ms'], 'k1_count', ks['k1_count']['avg_ms'], 'k4a', ks['k4a_simple']['avg_ms'])
PY
}
run "default" A=1
run "tail low" PJB_TAIL_LOW=1
run "default" A=1
run "tail low" PJB_TAIL_LOW=1
EXTRA="--queue 4" run "tail low, queue 4" PJB_TAIL_LOW=1
EXTRA="--queue 2" run "tail low, queue 2" PJB_TAIL_LOW=1
