# (experiment, round 6) what bounds the program between "context ready" and its last chain?  On the bench's prepared BAM (made by the first
# step): the copy paths alone (tools/debug/h2d_paths.cc on the real file), then the program with every host event logged, then a few settings
# of the transfer knobs, alternating.
OUT=gpurun_out; mkdir -p $OUT
BAM=/tmp/pjb_bench_e2e/prep/portcullis.sorted.alignments.bam
PJB_BENCH_E2E_REPS=3 PJB_BENCH_E2E_EARLY_REPS=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench e2e runs', d['e2e'].get('runs_s'), d['e2e'].get('error'))"
ls -la $BAM; nproc; numactl -H 2>/dev/null | head -5; cat /sys/fs/cgroup/cpu.max 2>/dev/null
/opt/rocm/bin/hipcc -O2 -o /tmp/h2d_paths tools/debug/h2d_paths.cc -lpthread
/tmp/h2d_paths $BAM 30 quick 2>&1 | tee $OUT/r06_h2d_paths_bam.txt
EXE=portcullis_amd/host/portcullis_amd
( PJB_PROFILE_HOST=2 $EXE junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/ev /tmp/pjb_bench_e2e/prep ) > $OUT/r06_e2e_events.txt 2>&1
grep -c "host event" $OUT/r06_e2e_events.txt
for k in 1 2 3 4; do for cfg in "2:2" "3:2" "4:2" "4:1" "6:1" "8:1" "0:1"; do
  s=$(date +%s.%N); PORTCULLIS_TRANSFER_SLOTS=${cfg%%:*} PORTCULLIS_READ_THREADS=${cfg##*:} PORTCULLIS_PINNED_BUFFERS=${NBUF:-12} $EXE junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/io /tmp/pjb_bench_e2e/prep > /dev/null 2>&1; e=$(date +%s.%N)
  python3 -c "print('e2e slots:threads $cfg: %.3f s' % ($e - $s))"; done; done | tee $OUT/r06_e2e_io_knobs.txt
python3 - $OUT/r06_e2e_io_knobs.txt <<'PY' | tee -a $OUT/r06_e2e_io_knobs.txt
import sys, collections, statistics
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    if ln.startswith("e2e "):
        k, v = ln[4:].split(": ")
        d[k].append(float(v.split()[0]))
for k, v in d.items():
    print(f"median {k}: {statistics.median(v):.3f} s  (min {min(v):.3f}, max {max(v):.3f}, {len(v)} runs)")
PY
