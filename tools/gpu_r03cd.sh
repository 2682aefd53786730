#!/bin/bash
# does block-parallel inflate scale on the box's cores?  (the reader's loop in isolation)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
g++ -O2 -std=c++17 -Iportcullis_amd/host/include -o /tmp/fi_speed tests/cpp/fast_inflate_check.cc portcullis_amd/host/src/fast_inflate.cc -lz -lpthread
for t in 1 4 8 16 32; do /tmp/fi_speed speed 512 $t; done
nproc; lscpu | grep -i "model name\|socket\|numa\|thread(s) per core" | head -8
