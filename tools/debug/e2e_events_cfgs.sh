# (experiment, round 6) event timelines of the program at several reader settings + alternating walls
OUT=gpurun_out; mkdir -p $OUT
PJB_BENCH_E2E_REPS=1 PJB_BENCH_E2E_EARLY_REPS=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench e2e runs', d['e2e'].get('runs_s'), d['e2e'].get('error'))"
EXE=portcullis_amd/host/portcullis_amd
for cfg in ${CFGS:-2:2 2:4 2:6 3:3}; do
  ( PJB_PROFILE_HOST=2 PORTCULLIS_TRANSFER_SLOTS=${cfg%%:*} PORTCULLIS_READ_THREADS=${cfg##*:} $EXE junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/ev /tmp/pjb_bench_e2e/prep ) > $OUT/r06_e2e_events_${cfg/:/_}.txt 2>&1
done
for k in 1 2 3 4 5 6 7 8; do for cfg in ${CFGS:-2:2 2:4 2:6 3:3}; do
  s=$(date +%s.%N); PORTCULLIS_TRANSFER_SLOTS=${cfg%%:*} PORTCULLIS_READ_THREADS=${cfg##*:} $EXE junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/io /tmp/pjb_bench_e2e/prep > /dev/null 2>&1; e=$(date +%s.%N)
  python3 -c "print('e2e slots:threads $cfg: %.3f s' % ($e - $s))"; done; done | tee $OUT/r06_e2e_io_knobs2.txt
python3 - $OUT/r06_e2e_io_knobs2.txt <<'PY' | tee -a $OUT/r06_e2e_io_knobs2.txt
import sys, collections, statistics
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    if ln.startswith("e2e "):
        k, v = ln[4:].split(": ")
        d[k].append(float(v.split()[0]))
for k, v in d.items():
    print(f"median {k}: {statistics.median(v):.3f} s  (min {min(v):.3f}, max {max(v):.3f}, {len(v)} runs)")
PY
