#!/bin/bash
# round 2, last tree: the whole GPU test suite, the profile round (rocprofv3 stats + PMC traffic + the bench line with its
# CPU baseline and end-to-end leg), then the end-to-end program's own event log, GPU timeline and inflate Gantt chart
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02g}
COMMIT=${2:-unknown}
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/${TAG}_pytest.log
export PJB_BENCH_KEEP_WORKDIR=1
bash tools/profile_round.sh $TAG $COMMIT
cd $GRAFT_REPO_ROOT
python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_C3.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('frac_alone'), d['roofline']['step_frac'], d['cpu_baseline']['value'], d['e2e'])"
W=/tmp/pjb_bench_e2e
PJB_PROFILE_HOST=2 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2> gpurun_out/${TAG}_e2e_host_events.txt
grep -E "device thread|workers|main:|context ready" gpurun_out/${TAG}_e2e_host_events.txt
rm -rf /tmp/e2e_prof
PJB_NORMAL_EXIT=1 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/e2e_prof -- portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc3 $W/prep > /dev/null 2> gpurun_out/${TAG}_e2e_rocprof.err
python tools/debug/e2e_timeline.py /tmp/e2e_prof > gpurun_out/${TAG}_e2e_gpu_timeline.txt 2>&1
python tools/debug/inflate_gantt.py /tmp/e2e_prof > gpurun_out/${TAG}_e2e_inflate_gantt.txt 2>&1
python3 tools/summarize_rocprof.py $(find /tmp/e2e_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_e2e_kernel_stats.csv
head -12 gpurun_out/${TAG}_e2e_gpu_timeline.txt
rm -rf $W
