#!/bin/bash
# bamfilt: where the wall clock goes at HEAD (PORTCULLIS_PROFILE marks of reader and writer)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_bamfilt_program.py --runs 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('runs', sorted(d['wall_s']))"
wd=/tmp/pjb_bamfilt
for k in 1 2; do
PORTCULLIS_PROFILE=1 portcullis_amd/host/portcullis_amd bamfilt -o $wd/filt/filtered.bam -c HARD -t 16 $wd/pass.junctions.tab $wd/prep/portcullis.sorted.alignments.bam > gpurun_out/r03bk_profile_$k.txt 2>&1
done
cat gpurun_out/r03bk_profile_2.txt | tail -80
