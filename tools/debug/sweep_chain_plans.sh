# (experiment) the step under different chain plans (PJB_BENCH_CHAINS: the chains and their order as given)
for plan in "" "0-5;6-13;14-24" "0-6;7-15;16-24" "0-7;8-17;18-24" "" "0-5;6-13;14-24" "0-6;7-15;16-24" "0-14;15-24"; do
  echo "== plan '$plan'"
  PJB_BENCH_CHAINS="$plan" python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['launches_per_step'], d['overlap_factor'])"
done
