#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
head -c 4G /dev/zero > /tmp/zeros.bin
cat /tmp/zeros.bin > /dev/null
for t in 4; do timeout 300 tools/debug/register_probe /tmp/zeros.bin $t; done 2>&1 | tee gpurun_out/r03ap2_register_probe.txt
