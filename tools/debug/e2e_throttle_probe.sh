# (experiment, round 6) is the program's "stands still for seconds" state (more targets in transfer, more readers) the container's CPU quota?
# cpu.stat of the cgroup before and after every run: usage_usec, nr_throttled, throttled_usec.
OUT=gpurun_out; mkdir -p $OUT
PJB_BENCH_E2E_REPS=1 PJB_BENCH_E2E_EARLY_REPS=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench e2e runs', d['e2e'].get('runs_s'), d['e2e'].get('error'))"
cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpu.stat | head -8
EXE=portcullis_amd/host/portcullis_amd
stat() { python3 -c "
d=dict(l.split() for l in open('/sys/fs/cgroup/cpu.stat'))
print(d.get('usage_usec',0), d.get('nr_throttled',0), d.get('throttled_usec',0))"; }
for k in 1 2 3; do for cfg in "2:2" "2:4" "3:2" "4:2" "0:1" ${EXTRA_CFGS:-}; do
  read u0 n0 t0 <<< "$(stat)"
  s=$(date +%s.%N); env ${ENVX:-PJB_NONE=1} PORTCULLIS_TRANSFER_SLOTS=${cfg%%:*} PORTCULLIS_READ_THREADS=${cfg##*:} $EXE junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/io /tmp/pjb_bench_e2e/prep > /dev/null 2>&1; e=$(date +%s.%N)
  read u1 n1 t1 <<< "$(stat)"
  python3 -c "print('e2e slots:threads $cfg: %.3f s   cpu %.2f s, throttled %d periods, %.2f s' % ($e - $s, ($u1-$u0)/1e6, $n1-$n0, ($t1-$t0)/1e6))"; done; done | tee $OUT/r06_e2e_throttle.txt
