#!/bin/bash
# run under gpurun: SQ counters per kernel (one pass per group), summarised as per-launch averages
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcsq_$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/bench_$i.log 2>&1 || true
  cp $(find /tmp/pmcsq_$i -name "*counter_collection.csv" | head -1) /tmp/pmcsq_$i.csv 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob('/tmp/pmcsq_*.csv')):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0].replace('pjb::', '')
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
names = ['k1_walk', 'k1_count', 'k1_emit', 'k4a_simple', 'k4b_generic', 'k4_pairs', 'k3_anchors_frag', 'kd_unique', 'rs_scatter', 'rs_hist', 'k5_frag_reduce']
ctrs = sorted({c for n in acc for c in acc[n]})
print('counter'.ljust(32) + ''.join(n[:13].rjust(14) for n in names))
for c in ctrs:
    row = c.ljust(32)
    for n in names:
        v = acc[n].get(c)
        row += (f'{sum(v) / len(v):14.3g}' if v else ' ' * 14)
    print(row)
PY
