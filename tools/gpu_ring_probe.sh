#!/bin/bash
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/ring_probe tools/debug/ring_probe.cc -lpthread 2>&1 | grep -v warning | head -5
F=/tmp/ring_probe.dat
/tmp/ring_probe $F 8 4 4 1 64 0 0 0 0
/tmp/ring_probe $F 8 4 4 1 64 0 0 0 0
/tmp/ring_probe $F 8 4 4 1 64 0 0 1 0
/tmp/ring_probe $F 8 4 4 1 64 0 0 2 0
/tmp/ring_probe $F 8 4 4 1 64 0 0 0 1
/tmp/ring_probe $F 8 4 4 0 64 0 0 1 0
/tmp/ring_probe $F 8 4 4 0 64 0 0 0 1
rm -f $F
