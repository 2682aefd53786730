#!/bin/bash
# bamfilt with the junction file read beside the first piece: A/B against the previous build on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/portcullis_amd/csrc:$LD_LIBRARY_PATH
for k in 1 2; do
python tools/bench_bamfilt_program.py --runs 7 --exe portcullis_amd/host/ab_old/portcullis_amd 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('old', sorted(d['wall_s']), d['kept_bytes_md5'])"
python tools/bench_bamfilt_program.py --runs 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('new', sorted(d['wall_s']), d['kept_bytes_md5'])"
done
