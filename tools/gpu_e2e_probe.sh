#!/bin/bash
# the end-to-end leg's files, then the ring probe on the real BAM, then the program
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 -o /tmp/ring_probe tools/debug/ring_probe.cc -lpthread 2>&1 | grep -v warning | head -5
PJB_BENCH_KEEP_WORKDIR=1 PJB_BENCH_E2E_REPS=1 timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/e2e_tr_bench.json 2> gpurun_out/e2e_tr_bench.err
python -c "import json; print(json.load(open('gpurun_out/e2e_tr_bench.json'))['e2e']['runs_s'])"
W=/tmp/pjb_bench_e2e
B=$W/prep/portcullis.sorted.alignments.bam
ls -la $B; df -h /tmp | tail -1; free -g | head -2; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/memory.current 2>/dev/null
PJB_READSPAN_DEBUG=1 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2> gpurun_out/e2e_readspan.txt
grep -c readSpan gpurun_out/e2e_readspan.txt; grep readSpan gpurun_out/e2e_readspan.txt | awk 'NR%40==1' | head -14
PORTCULLIS_TRANSFER_SLOTS=1 PJB_READSPAN_DEBUG=1 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2> gpurun_out/e2e_readspan1.txt; grep readSpan gpurun_out/e2e_readspan1.txt | awk 'NR%40==1' | head -14
