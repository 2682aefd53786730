# (experiment) the step under different chain plans: a small last chain leaves less of its tail without a K1 stage beside it
for plan in "" "0-3;4-9;10-18;19-24" "0-3;4-9;10-21;22-24" "0-3;4-9;10-16;17-24" "19-24;0-3;4-9;10-18" "0-4;5-11;12-21;22-24"; do
  echo "== plan '$plan'"
  PJB_BENCH_CHAINS="$plan" python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['launches_per_step'], d['overlap_factor'])"
done
