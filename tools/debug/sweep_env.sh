# (experiment) the step under values of one environment variable: bash tools/debug/sweep_env.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['launches_per_step'], d['overlap_factor'])
for k in d['kernels']:
    if k['name'] in ('kd_ends','kd_table','kd_reset'): print('  ', k['name'], round(k['avg_ms']*1000,1))"
done
