#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
PJB_BENCH_KEEP_WORKDIR=1 PJB_BENCH_E2E_REPS=1 timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/e2e_tr_bench.json 2> gpurun_out/e2e_tr_bench.err
W=/tmp/pjb_bench_e2e
sync
for i in 1 2; do portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2>&1; done
export PJB_NORMAL_EXIT=1
for ctx in 2 4; do
rm -rf /tmp/e2e_prof
PORTCULLIS_CTX_PER_GPU=$ctx rocprofv3 --kernel-trace --output-format csv -d /tmp/e2e_prof -- portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc3 $W/prep > /dev/null 2> gpurun_out/e2e_tr_rocprof.err
echo "== contexts per GPU: $ctx"
python tools/debug/inflate_gantt.py /tmp/e2e_prof | tee gpurun_out/inflate_gantt_$ctx.txt
done
