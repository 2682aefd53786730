# (experiment) the step under the K1 scheduling switches of the library, with the current kernels (bench.py quick mode)
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-back-to-back 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['name']:x for x in d['kernels']}
print(round(d['ms_per_step'],3), 'emit in-step us', round(1e3*k['k1_emit']['avg_ms'],1), 'overlap', d['overlap_factor'], 'kd_table', round(1e3*k['kd_table']['avg_ms'],1), 'kd_ends', round(1e3*k['kd_ends']['avg_ms'],1))"; }
for t in ${SWEEP:-4 8 12 16 24 32 64 8 4}; do run PJB_K1E_TILES=$t; done
