#!/bin/bash
# pjb_create: kernel attributes (code object load) beside the stream creation
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_bamfilt.py -x -q 2>&1 | tail -2
python - <<'PY'
import sys, time, os
sys.path.insert(0, '.')
from portcullis_amd import ffi
for env in ('1', None, '1', None):
    if env: os.environ['PJB_CREATE_SERIAL'] = env
    else: os.environ.pop('PJB_CREATE_SERIAL', None)
    ts = []
    for k in range(5):
        t = time.perf_counter()
        ctx = ffi.Context(0, "UNKNOWN")
        ts.append((time.perf_counter() - t) * 1e3)
        ctx.close() if hasattr(ctx, 'close') else ctx.__exit__(None, None, None)
    print('serial' if env else 'beside', [round(x, 1) for x in ts])
PY
for k in 1 2; do
python tools/bench_bamfilt_program.py --runs 7 --env PJB_CREATE_SERIAL=1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bamfilt serial', sorted(d['wall_s']))"
python tools/bench_bamfilt_program.py --runs 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bamfilt beside', sorted(d['wall_s']))"
done
wd=/tmp/pjb_bamfilt
PJB_CREATE_TRACE=1 portcullis_amd/host/portcullis_amd bamfilt -o $wd/filt/f2.bam -c HARD -t 16 $wd/pass.junctions.tab $wd/prep/portcullis.sorted.alignments.bam 2>&1 | grep pjb_create
