run() { echo "== rank $1 gb $2"; PJB_BENCH_AS_RANK=$1 python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-back-to-back --group-bases $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['launches_per_step'], d['overlap_factor'], d['config']['chains'], d['config'].get('targets_rank'), d['config']['reads_rank0'])
for k in d['kernels'][:12]: print('    ',k['name'],k['avg_ms'],k['launches_per_step'])"; }
run 0/8 1073741824; run 0/8 300000000; run 0/8 200000000; run 0/8 150000000
run 0/4 1073741824; run 0/4 420000000; run 0/4 300000000
