# (experiment, round 6) the program on the bench's prepared BAM under several environments, alternating, host timers on:
#   ENVS="A=1;B=2 C=3;..." (';'-separated settings, each a space-separated list of VAR=value; "-" = nothing), REPS (default 6)
OUT=gpurun_out; mkdir -p $OUT
NAME=${NAME:-env_ab}
PJB_BENCH_E2E_REPS=1 PJB_BENCH_E2E_EARLY_REPS=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench e2e runs', d['e2e'].get('runs_s'), d['e2e'].get('error'))"
EXE=${EXE:-portcullis_amd/host/portcullis_amd}
IFS=';' read -ra SETTINGS <<< "${ENVS:--}"
for k in $(seq 1 ${REPS:-6}); do for i in "${!SETTINGS[@]}"; do
  cfg="${SETTINGS[$i]}"; [ "$cfg" = "-" ] && cfg="PJB_NONE=1"
  sleep ${PAUSE:-3}  # (a run right behind another one waits for the driver to take that one's device memory apart: profiles/r06_e2e_pause.txt)
  exe=$EXE; for kv in $cfg; do case $kv in EXE=*) exe=${kv#EXE=};; esac; done  # (a setting may name another build of the program: EXE=path)
  s=$(date +%s.%N); env $cfg PJB_PROFILE_HOST=1 $exe junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/ab$i /tmp/pjb_bench_e2e/prep > /tmp/ab_$i.txt 2>&1; e=$(date +%s.%N)
  python3 - "$cfg" $s $e /tmp/ab_$i.txt <<'PY'
import sys, re
cfg, s, e, f = sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
t = open(f).read()
g = lambda pat: (re.search(pat, t) or [None, '?'])[1]
print('e2e [%s]: %.3f s | ready %s, workers done %s, outputs written %s, workers %s | dev: idle %s GENOME %s BAM %s BAMEND %s FINISH %s collect %s' % (
    cfg, e - s, g(r't=([\d.]+) s: device thread: context ready'), g(r't=([\d.]+) s: workers and device threads done'), g(r't=([\d.]+) s: outputs written'), g(r'workers ([\d.]+) s,'),
    g(r'idle ([\d.]+),'), g(r'GENOME ([\d.]+),'), g(r'BAM ([\d.]+),'), g(r'BAMEND ([\d.]+),'), g(r'FINISH ([\d.]+),'), g(r'collect ([\d.]+),')))
PY
done; done | tee $OUT/r06_e2e_$NAME.txt
md5sum /tmp/pjb_bench_e2e/prof/ab*.junctions.tab | tee -a $OUT/r06_e2e_$NAME.txt
python3 - $OUT/r06_e2e_$NAME.txt <<'PY' | tee -a $OUT/r06_e2e_$NAME.txt
import sys, collections, statistics, re
d = collections.defaultdict(list); w = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m = re.match(r'e2e \[(.*?)\]: ([\d.]+) s .*workers ([\d.?]+) \|', ln)
    if m:
        d[m[1]].append(float(m[2]))
        if m[3] != '?': w[m[1]].append(float(m[3]))
for k, v in d.items():
    print(f"median [{k}]: wall {statistics.median(v):.3f} s (min {min(v):.3f}, max {max(v):.3f}, {len(v)} runs); workers phase {statistics.median(w[k]) if w[k] else float('nan'):.3f} s (min {min(w[k]) if w[k] else 0:.3f})")
PY
