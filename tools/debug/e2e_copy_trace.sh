# (experiment, round 6) the copies of one run of the program (rocprofv3 --memory-copy-trace beside the kernel trace): is the DMA busy or waiting?
OUT=gpurun_out; mkdir -p $OUT
PJB_BENCH_E2E_REPS=1 PJB_BENCH_E2E_EARLY_REPS=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench e2e runs', d['e2e'].get('runs_s'), d['e2e'].get('error'))"
sleep 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/e2e_trace
s=$(date +%s.%N)
PJB_NORMAL_EXIT=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/e2e_trace -- $R/portcullis_amd/host/portcullis_amd junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/kt /tmp/pjb_bench_e2e/prep > /tmp/e2e_trace_stdout.txt 2>&1
e=$(date +%s.%N); python3 -c "print('traced run: %.3f s' % ($e - $s))"
tail -5 /tmp/e2e_trace_stdout.txt; f=$(find /tmp/e2e_trace -name "*kernel_trace.csv" | head -1); ls -la $f
cp $f $R/$OUT/r06_e2e_copy_trace_kernels.csv; m=$(find /tmp/e2e_trace -name "*memory_copy_trace.csv" | head -1); ls -la $m; cp $m $R/$OUT/r06_e2e_copy_trace_copies.csv
python3 - $f <<'PY' | tee $R/$OUT/r06_e2e_copy_trace.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]) for r in rows]
t0 = min(a for a, b, n in ev); t1 = max(b for a, b, n in ev)
print('kernels %d, first to last %.3f s' % (len(ev), (t1 - t0) / 1e9))
def union(iv):
    iv = sorted(iv); tot = 0
    if not iv: return 0
    ca, cb = iv[0]
    for a, b in iv[1:]:
        if a > cb: tot += cb - ca; ca, cb = a, b
        else: cb = max(cb, b)
    return tot + cb - ca
print('any kernel running: %.3f s' % (union([(a, b) for a, b, n in ev]) / 1e9))
by = collections.defaultdict(list)
for a, b, n in ev: by[n].append((a, b))
print('%-28s %6s %10s %10s' % ('kernel', 'calls', 'sum s', 'union s'))
for n, iv in sorted(by.items(), key=lambda kv: -sum(b - a for a, b in kv[1]))[:16]:
    print('%-28s %6d %10.3f %10.3f' % (n[:28], len(iv), sum(b - a for a, b in iv) / 1e9, union(iv) / 1e9))
inf = [(a, b) for a, b, n in ev if n.startswith('bgzf')]
print('bgzf_* union %.3f s; everything else union %.3f s' % (union(inf) / 1e9, union([(a, b) for a, b, n in ev if not n.startswith('bgzf')]) / 1e9))
# 50 ms bins: fraction covered by any kernel / by inflate
bins = int((t1 - t0) / 5e7) + 1
for k in range(bins):
    lo, hi = t0 + k * 5e7, t0 + (k + 1) * 5e7
    c_any = union([(max(a, lo), min(b, hi)) for a, b, n in ev if a < hi and b > lo] or [(lo, lo)])
    c_inf = union([(max(a, lo), min(b, hi)) for a, b in inf if a < hi and b > lo] or [(lo, lo)])
    print('t %.2f s: any %3.0f %%  inflate %3.0f %%' % (k * 0.05, 100 * c_any / 5e7, 100 * c_inf / 5e7))
PY
