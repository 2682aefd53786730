#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tests/e2e_bench.py --config C2 --threads 16 --workdir /tmp/pjb_e2e_c2 --keep --no-oracle --repeat 1 > /dev/null 2>&1
W=/tmp/pjb_e2e_c2
for v in A=1 PJB_SERIAL_SLOT_INIT=1 A=2 PJB_SERIAL_SLOT_INIT=1; do
  env $v PJB_PROFILE_HOST=1 portcullis_amd/host/portcullis_amd junc -t 16 -o $W/out/x $W/prep 2>&1 | grep -E "pjb_create|context ready|main:" | tr '\n' ' '; echo " [$v]"
done 2>&1 | tee gpurun_out/r03av.txt
