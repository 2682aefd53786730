#!/bin/bash
# differential campaigns at the round's last tree (new seeds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python tests/fuzz_campaign.py --seeds 500 --start 91000 2>&1 | tail -2
  timeout 1500 python tests/fuzz_groups.py --seeds 120 --start 92000 2>&1 | tail -2
  timeout 1500 python tests/fuzz_extra.py --seeds 80 --start 93000 2>&1 | tail -2 ) | tee gpurun_out/r03cj_fuzz_campaigns.txt
