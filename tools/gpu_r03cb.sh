#!/bin/bash
# --extra: where the 1.6 x stands at this tree (marks, XTRACE of the host side, kernel table)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in 1 2 3; do python tools/bench_extra.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['plain_ms'], d['extra_ms'], d['extra_over_plain'])"; done
python tools/debug/extra_marks.py 2>&1 | tail -2
PJB_XTRACE=1 python tools/debug/extra_marks.py 2>&1 | grep -v "^plain\|^extra" | tail -40
python tools/debug/extra_breakdown.py 2>&1 | grep -E "kx_|k1_count|marks" | head -30
