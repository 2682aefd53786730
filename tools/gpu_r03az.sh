#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_bamfilt_program.py --runs 3 | cut -c1-330
python tools/bench_bamfilt_program.py --runs 3 --env PORTCULLIS_PINNED_BUFFERS=1 | cut -c1-330
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03az_bench.json 2> gpurun_out/r03az_bench.err
W=/tmp/pjb_bench_e2e
for k in 1 2 3; do
( time PJB_PROFILE_HOST=1 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null ) 2>&1 | grep -E "host profile|real" | grep -v "main entered\|leaving"
done 2>&1 | tee gpurun_out/r03az_host.txt
( time portcullis_amd/host/portcullis_amd junc --help > /dev/null ) 2>&1 | grep real
