# (experiment) chains in flight (bench.py --queue)
for q in 3 2 4 3; do
  echo "== --queue $q"
  python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --queue $q 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['back_to_back']['ms_per_pass'])"
done
