#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python tests/fuzz_campaign.py --seeds 150 --start 31000 2>&1 | tail -6 ) | tee gpurun_out/r03bd_fuzz_campaign.txt
( timeout 1200 python tests/fuzz_groups.py --seeds 100 --start 41000 2>&1 | tail -6 ) | tee gpurun_out/r03bd_fuzz_groups.txt
