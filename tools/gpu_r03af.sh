#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_bamfilt_program.py --runs 1 > /dev/null 2>&1
W=/tmp/pjb_bamfilt
for v in A=1 A=2; do
( time env $v PORTCULLIS_PROFILE=1 portcullis_amd/host/portcullis_amd bamfilt -o $W/filt/f.bam -c HARD -t 16 $W/pass.junctions.tab $W/prep/portcullis.sorted.alignments.bam ) 2>&1 | grep -E "profile|real|Total"
done
