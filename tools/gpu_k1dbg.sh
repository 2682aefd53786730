#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for dbg in 0 1 2 3; do
  PJB_K1W_DEBUG=$dbg timeout 600 python tools/debug/k1_alone.py 2>&1 | tail -3
done
