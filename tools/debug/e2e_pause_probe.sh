# (experiment, round 6) does the program's slow state (a 1.5 - 2 s stall of the copies somewhere in the run) come from the process before it --
# the driver clearing that process's 100+ GB of device memory?  Runs back to back against runs with a pause before them.
OUT=gpurun_out; mkdir -p $OUT
PJB_BENCH_E2E_REPS=1 PJB_BENCH_E2E_EARLY_REPS=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench e2e runs', d['e2e'].get('runs_s'), d['e2e'].get('error'))"
EXE=portcullis_amd/host/portcullis_amd
one() { s=$(date +%s.%N); $EXE junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/pp /tmp/pjb_bench_e2e/prep > /dev/null 2>&1; e=$(date +%s.%N); python3 -c "print('e2e $1: %.3f s' % ($e - $s))"; }
for round in 1 2 3; do
  for k in 1 2 3 4 5 6; do one "back to back"; done
  for k in 1 2 3 4 5 6; do sleep ${PAUSE:-3}; one "after a pause"; done
done | tee $OUT/r06_e2e_pause.txt
python3 - $OUT/r06_e2e_pause.txt <<'PY' | tee -a $OUT/r06_e2e_pause.txt
import sys, collections, statistics
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    if ln.startswith("e2e "):
        k, v = ln[4:].split(": ")
        d[k].append(float(v.split()[0]))
for k, v in d.items():
    print(f"{k}: median {statistics.median(v):.3f} s, min {min(v):.3f}, max {max(v):.3f}, runs over 2.4 s: {sum(x > 2.4 for x in v)} of {len(v)}")
PY
