#!/bin/bash
# bamfilt: input blocks inflated on the device, page-locked buffers by default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bamfilt.py -x -q 2>&1 | tail -3
python tools/bench_bamfilt_program.py --runs 5 --env PORTCULLIS_HOST_INFLATE=1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('zlib inflate  ', sorted(d['wall_s']), d['kept_bytes_md5'])"
python tools/bench_bamfilt_program.py --runs 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('device inflate', sorted(d['wall_s']), d['kept_bytes_md5'])"
wd=/tmp/pjb_bamfilt
for k in 1 2; do
PORTCULLIS_PROFILE=1 portcullis_amd/host/portcullis_amd bamfilt -o $wd/filt/filtered.bam -c HARD -t 16 $wd/pass.junctions.tab $wd/prep/portcullis.sorted.alignments.bam > gpurun_out/r03bn_profile_$k.txt 2>&1
done
grep "profile\|pjb_create" gpurun_out/r03bn_profile_2.txt
