#!/bin/bash
# tests that exercise the host pipeline, then the end-to-end leg, repeats of the program, the inflate Gantt chart
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
PJB_BENCH_KEEP_WORKDIR=1 PJB_BENCH_E2E_REPS=1 timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/e2e_tr_bench.json 2> gpurun_out/e2e_tr_bench.err
python -c "import json; print(json.load(open('gpurun_out/e2e_tr_bench.json'))['e2e'])"
W=/tmp/pjb_bench_e2e
sync
run() {
  local t0=$EPOCHREALTIME
  env PJB_PROFILE_HOST=1 "$@" portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2> /tmp/err.txt
  local t1=$EPOCHREALTIME
  python3 - "$*" $t0 $t1 <<'PY'
import re, sys
err = open('/tmp/err.txt').read()
t0, t1 = float(sys.argv[2]), float(sys.argv[3])
ent = float(re.search(r"main entered at epoch ([0-9.]+)", err).group(1))
lea = float(re.search(r"leaving main at epoch ([0-9.]+)", err).group(1))
print(f"{sys.argv[1]}: wall {t1 - t0:.2f} s = {ent - t0:.2f} before main + {lea - ent:.2f} in main + {t1 - lea:.2f} after")
PY
}
run A=1; for i in 1 2 3 4; do run A=1; run PJB_INFLATE_WG_PER_CU=3; run PJB_INFLATE_LOW_PRIORITY=1; run PJB_INFLATE_WG_PER_CU=3 PJB_INFLATE_LOW_PRIORITY=1; done; md5sum $W/out/pc2.junctions.tab
PJB_PROFILE_HOST=2 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2> gpurun_out/e2e_tr_host.txt
grep -E "device thread|workers|main:|context ready" gpurun_out/e2e_tr_host.txt
rm -rf /tmp/e2e_prof
export PJB_NORMAL_EXIT=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/e2e_prof -- portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc3 $W/prep > /dev/null 2> gpurun_out/e2e_tr_rocprof.err
python tools/debug/inflate_gantt.py /tmp/e2e_prof | tee gpurun_out/inflate_gantt.txt
python tools/debug/e2e_timeline.py /tmp/e2e_prof > gpurun_out/e2e_timeline.txt 2>&1
sed -n 2,4p gpurun_out/e2e_timeline.txt
