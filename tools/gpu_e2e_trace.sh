#!/bin/bash
# the end-to-end leg's files, then the program under rocprofv3 (kernel + memory-copy trace) and with the host timers
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
PJB_BENCH_KEEP_WORKDIR=1 PJB_BENCH_E2E_REPS=${REPS:-1} timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/e2e_tr_bench.json 2> gpurun_out/e2e_tr_bench.err
python -c "import json; print(json.load(open('gpurun_out/e2e_tr_bench.json'))['e2e'])"
W=/tmp/pjb_bench_e2e
PJB_PROFILE_HOST=2 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2> gpurun_out/e2e_tr_host.txt
grep -E "device thread|workers|main:|context ready" gpurun_out/e2e_tr_host.txt
grep -c "host event" gpurun_out/e2e_tr_host.txt
rm -rf /tmp/e2e_prof
export PJB_NORMAL_EXIT=1
( time rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/e2e_prof -- portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc3 $W/prep > /dev/null 2> gpurun_out/e2e_tr_rocprof.err ) 2>&1 | tail -3
python tools/debug/e2e_timeline.py /tmp/e2e_prof > gpurun_out/e2e_timeline.txt 2>&1
head -70 gpurun_out/e2e_timeline.txt
python tools/debug/inflate_gantt.py /tmp/e2e_prof > gpurun_out/e2e_inflate_gantt.txt 2>&1
tail -45 gpurun_out/e2e_inflate_gantt.txt
