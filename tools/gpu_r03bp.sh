#!/bin/bash
# bamfilt: the scan two pieces ahead of the decisions; contexts without chain slots
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bamfilt.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -3
for k in 1 2; do
python tools/bench_bamfilt_program.py --runs 5 --env PORTCULLIS_SCAN_AHEAD=0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('in turns ', sorted(d['wall_s']), d['kept_bytes_md5'])"
python tools/bench_bamfilt_program.py --runs 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ahead 2  ', sorted(d['wall_s']), d['kept_bytes_md5'])"
done
python tools/bench_bamfilt_program.py --runs 5 --env PORTCULLIS_SCAN_AHEAD=1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ahead 1  ', sorted(d['wall_s']), d['kept_bytes_md5'])"
python tools/bench_bamfilt_program.py --runs 5 --env PORTCULLIS_SCAN_AHEAD=3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ahead 3  ', sorted(d['wall_s']), d['kept_bytes_md5'])"
wd=/tmp/pjb_bamfilt
for k in 1 2; do
PJB_CREATE_TRACE=1 PORTCULLIS_PROFILE=1 portcullis_amd/host/portcullis_amd bamfilt -o $wd/filt/filtered.bam -c HARD -t 16 $wd/pass.junctions.tab $wd/prep/portcullis.sorted.alignments.bam > gpurun_out/r03bp_profile_$k.txt 2>&1
done
grep "profile\|pjb_create" gpurun_out/r03bp_profile_2.txt
