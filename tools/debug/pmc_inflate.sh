#!/bin/bash
# run under gpurun: SQ / LDS counters of the inflate kernels (bgzf_decode, bgzf_resolve; PJB_INFLATE_V1=1: bgzf_inflate) on the
# C2 BAM (one launch of 64 k blocks).  Output: gpurun_out/pmc_inflate/summary.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_inflate
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" \
           "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmci_$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_inflate.py --times 2 --chunk-mb 16384 > $OUT/bench_$i.log 2>&1 || true
  cp $(find /tmp/pmci_$i -name "*counter_collection.csv" | head -1) /tmp/pmci_$i.csv 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob('/tmp/pmci_*.csv')):
    for r in csv.DictReader(open(f)):
        for k in ('bgzf_inflate', 'bgzf_decode', 'bgzf_resolve'):
            if k in r['Kernel_Name']:
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print('----', k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f'{c:32s} {max(v):14.4g}   (launches {len(v)})')
PY
