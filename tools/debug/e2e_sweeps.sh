# (experiment, round 5) the program's host-side knobs on the bench's prepared BAM (made by the first call, kept in /tmp for the rest)
run() { echo "== $1 ${3:-}"; PJB_BENCH_E2E_SWEEP="$1" PJB_BENCH_E2E_REPS=$2 PJB_BENCH_E2E_EARLY_REPS=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['e2e'].get('runs_s'), d['e2e'].get('error'))"; }
run "PJB_DUMMY=0,0,0" 3
run "PORTCULLIS_TRANSFER_SLOTS=2,3,2,3,2,3,2,3,2,3" 10
export PORTCULLIS_TRANSFER_SLOTS=3
run "PORTCULLIS_READ_THREADS=1,2,3,1,2,3,1,2,3" 9 "(slots 3)"
run "PORTCULLIS_PINNED_BUFFERS=12,18,12,18,12,18" 6 "(slots 3)"
unset PORTCULLIS_TRANSFER_SLOTS
run "PORTCULLIS_TRANSFER_SLOTS=3,2,3,2,3,2,3,2,3,2" 10
