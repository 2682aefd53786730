#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bamfilt.py -x -q 2>&1 | tail -2
python tools/bench_bamfilt_program.py --runs 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bamfilt', sorted(d['wall_s']), d['kept_bytes_md5'])"
