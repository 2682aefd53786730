#!/bin/bash
# the scan alone: threads and placement
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/bench_bamfilt_program.py --runs 1 > /dev/null 2>&1
H=portcullis_amd/host
g++ -O2 -std=c++17 -I$H/include -Iinclude -o /tmp/scan_probe tools/debug/scan_probe.cc $H/src/bam_reader.cc $H/src/fast_inflate.cc $H/src/bam_writer.cc -lz -lpthread || exit 1
B=/tmp/pjb_bamfilt/prep/portcullis.sorted.alignments.bam
for t in 8 16 32; do /tmp/scan_probe $B $t | tail -1; done
echo "zlib:"; PORTCULLIS_ZLIB_INFLATE=1 /tmp/scan_probe $B 16 | tail -1
echo "node 0 only:"; taskset -c 0-63,128-191 /tmp/scan_probe $B 16 | tail -1
echo "node 1 only:"; taskset -c 64-127,192-255 /tmp/scan_probe $B 16 | tail -1
echo "16 physical cores of node 0:"; taskset -c 0-15 /tmp/scan_probe $B 16 | tail -1
PORTCULLIS_PROFILE_PIECES=1 /tmp/scan_probe $B 16 2>&1 | grep "scan piece" | tail -4
