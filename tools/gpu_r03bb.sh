#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bamfilt.py -x -q 2>&1 | tail -3
python tools/bench_bamfilt_program.py --runs 7 | tee gpurun_out/r03bb_bamfilt_program.json | cut -c1-330
W=/tmp/pjb_bamfilt
( time PORTCULLIS_PROFILE=1 PORTCULLIS_NO_FORK=1 portcullis_amd/host/portcullis_amd bamfilt -o $W/filt/f.bam -c HARD -t 16 $W/pass.junctions.tab $W/prep/portcullis.sorted.alignments.bam ) 2>&1 | grep -E "profile|real|Total"
