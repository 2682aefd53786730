#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in A=1 PJB_XPRE_LATE=1 A=2 PJB_XPRE_LATE=1; do
  env $v python tools/bench_extra.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', d['plain_ms'], d['extra_ms'], d['extra_over_plain'])"
done 2>&1 | tee gpurun_out/r03ab.txt
