#!/bin/bash
# bamfilt: contexts without chain slots; what pjb_create is made of (PJB_CREATE_TRACE)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_bamfilt_program.py --runs 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('runs', sorted(d['wall_s']), d['kept_bytes_md5'])"
wd=/tmp/pjb_bamfilt
for k in 1 2; do
PJB_CREATE_TRACE=1 PORTCULLIS_PROFILE=1 portcullis_amd/host/portcullis_amd bamfilt -o $wd/filt/filtered.bam -c HARD -t 16 $wd/pass.junctions.tab $wd/prep/portcullis.sorted.alignments.bam > gpurun_out/r03bl_profile_$k.txt 2>&1
done
grep "profile\|pjb_create" gpurun_out/r03bl_profile_2.txt
