#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
head -c 12G /dev/zero > /tmp/zeros.bin
cat /tmp/zeros.bin > /dev/null
for t in 2 4 6 8 16; do timeout 300 tools/debug/register_probe /tmp/zeros.bin $t | grep -v anonymous | grep -v hipHostMalloc; done 2>&1 | tee gpurun_out/r03ap3_register_probe.txt
