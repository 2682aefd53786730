#!/bin/bash
# (experiment) texture-addresser / L1 counters of the pjb kernels: is k1_emit's pace set by the number of cache lines its gathers touch?
# every group under its own time limit; the raw tables are summarised and deleted group by group (they are large)
OUT=$GRAFT_REPO_ROOT/gpurun_out/ta_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TA\|TCP\|TD\)_[A-Za-z0-9_]*" | sort -u > $OUT/avail.txt
wc -l $OUT/avail.txt
i=0
for grp in "${@:2}"; do
  rm -rf /tmp/ta_g
  timeout 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/ta_g -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/bench_g$i.log 2>&1 || echo "group $i ($grp) failed: $(tail -2 $OUT/bench_g$i.log)"
  python3 - /tmp/ta_g $OUT/summary_g$i.csv <<'PY'
import csv, glob, sys, collections
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(src + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'pjb::' not in k: continue
        k = k.split('pjb::')[1].split('<')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(dst, 'w') as fo:
    fo.write('kernel,' + ','.join(names) + '\n')
    for k in sorted(acc):
        fo.write(k + ',' + ','.join('%.0f' % (acc[k][c] / max(cnt[k][c], 1)) for c in names) + '\n')
print(open(dst).read())
PY
  du -sh /tmp/ta_g 2>/dev/null; rm -rf /tmp/ta_g
  i=$((i+1))
done
