#!/bin/bash
# Issue / stall counters of the pjb kernels (run under gpurun): one --pmc pass per counter group.
# usage: tools/pmc_sq.sh <tag>
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
# SQ_GROUPS=n: only the first n groups (instruction counts: 2)
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-back-to-back > $OUT/bench_g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
  if [ -n "$SQ_GROUPS" ] && [ $i -ge $SQ_GROUPS ]; then break; fi
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'pjb::' not in k: continue
        k = k.split('pjb::')[1].split('<')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(out + '/summary.csv', 'w') as fo:
    fo.write('kernel,' + ','.join(names) + '\n')
    for k in sorted(acc):
        fo.write(k + ',' + ','.join('%.0f' % (acc[k][c] / max(cnt[k][c], 1)) for c in names) + '\n')
print(open(out + '/summary.csv').read())
PY
rm -rf $OUT/g*   # (the raw counter tables stay on the box: gpurun copies back 64 MiB at most)
