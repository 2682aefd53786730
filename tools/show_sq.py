#!/usr/bin/env python3
"""Per-kernel view of tools/pmc_sq.sh's summary.csv (per-launch averages): waves, busy cycles, instruction mix per wave,
share of wave-cycles spent waiting / issuing."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))


def f(r, k):
    try:
        return float(r.get(k, 0) or 0)
    except ValueError:
        return 0.0


rows.sort(key=lambda r: -f(r, "SQ_BUSY_CYCLES"))
print(f"{'kernel':24s} {'waves':>9s} {'busy_cyc':>10s} {'valu/w':>8s} {'salu/w':>8s} {'vmrd/w':>7s} {'vmwr/w':>7s} {'lds/w':>7s} {'smem/w':>7s} "
      f"{'wait%':>6s} {'act_valu%':>9s} {'act_vmem%':>9s} {'act_lds%':>8s} {'waitlds%':>8s} {'ldsconf':>9s}")
for r in rows[:24]:
    w = max(f(r, "SQ_WAVES"), 1.0)
    wc = max(f(r, "SQ_WAVE_CYCLES"), 1.0)
    print(f"{r['kernel'][:24]:24s} {w:9.0f} {f(r, 'SQ_BUSY_CYCLES'):10.0f} {f(r, 'SQ_INSTS_VALU') / w:8.0f} {f(r, 'SQ_INSTS_SALU') / w:8.0f} "
          f"{f(r, 'SQ_INSTS_VMEM_RD') / w:7.1f} {f(r, 'SQ_INSTS_VMEM_WR') / w:7.1f} {f(r, 'SQ_INSTS_LDS') / w:7.1f} {f(r, 'SQ_INSTS_SMEM') / w:7.1f} "
          f"{100 * f(r, 'SQ_WAIT_INST_ANY') / wc:6.1f} {100 * f(r, 'SQ_ACTIVE_INST_VALU') / wc:9.1f} {100 * f(r, 'SQ_ACTIVE_INST_VMEM') / wc:9.1f} "
          f"{100 * f(r, 'SQ_ACTIVE_INST_LDS') / wc:8.1f} {100 * f(r, 'SQ_WAIT_INST_LDS') / wc:8.1f} {f(r, 'SQ_LDS_BANK_CONFLICT'):9.0f}")
