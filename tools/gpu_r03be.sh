#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python tests/fuzz_extra.py --seeds 80 --start 7000 2>&1 | tail -8 ) | tee gpurun_out/r03be_fuzz_extra.txt
( timeout 1500 python tests/fuzz_campaign.py --seeds 400 --start 52000 2>&1 | tail -4 ) | tee -a gpurun_out/r03be_fuzz_extra.txt
