#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
which perf strace gdb bpftrace numactl lscpu 2>&1 | head
lscpu | grep -E "NUMA|Socket|Model name|Thread" | head
PJB_BENCH_KEEP_WORKDIR=1 PJB_BENCH_E2E_REPS=1 timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/e2e_tr_bench.json 2> gpurun_out/e2e_tr_bench.err
python -c "import json; print(json.load(open('gpurun_out/e2e_tr_bench.json'))['e2e']['runs_s'])"
W=/tmp/pjb_bench_e2e
t0=$EPOCHREALTIME; sync; t1=$EPOCHREALTIME; python3 -c "print('sync %.2f s' % ($t1 - $t0))"
grep -E "Dirty|Writeback:" /proc/meminfo
for i in 1 2; do portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2>&1; done
t0=$EPOCHREALTIME
PJB_PROFILE_HOST=2 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2> gpurun_out/e2e_tr_host.txt
t1=$EPOCHREALTIME
python3 -c "print('wall %.2f s' % ($t1 - $t0))"
grep -E "device thread|workers|main:|context ready" gpurun_out/e2e_tr_host.txt
rm -rf /tmp/e2e_prof
export PJB_NORMAL_EXIT=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/e2e_prof -- portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc3 $W/prep > /dev/null 2> gpurun_out/e2e_tr_rocprof.err
python tools/debug/e2e_timeline.py /tmp/e2e_prof > gpurun_out/e2e_timeline.txt 2>&1
head -60 gpurun_out/e2e_timeline.txt
