# (experiment) one GPU with the share of rank 0 of N: the step under different chain sizes (bench.py --group-bases)
run() { echo "== rank $1 of $2, --group-bases $3"; PJB_BENCH_AS_RANK=$1/$2 python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-back-to-back --group-bases $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['launches_per_step'], d['overlap_factor'])"; }
run 0 2 1073741824; run 0 2 830000000; run 0 2 700000000
run 0 4 1073741824; run 0 4 420000000; run 0 4 330000000
run 3 4 1073741824; run 3 4 420000000
run 0 3 1073741824; run 0 3 560000000
