# (experiment) three chains with a smaller last one: the last chain's tail is the part of a step nothing runs beside
run() { echo "== ${1:-default}"; PJB_BENCH_CHAINS="$1" python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-back-to-back 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['config']['chains'], 'overlap', d['overlap_factor'])"; }
run ""
run "0-5;6-13;14-24"
run "0-6;7-14;15-24"
run "0-5;6-14;15-24"
run "0-5;6-12;13-24"
run ""
