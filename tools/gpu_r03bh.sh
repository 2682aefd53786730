#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extra.py -x -q 2>&1 | tail -3
for k in 1 2 3; do python tools/bench_extra.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['plain_ms'], d['extra_ms'], d['extra_over_plain'])"; done
python tools/debug/extra_breakdown.py 2>&1 | grep -E "kx_|marks"
