#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats.csv to the
pjb:: kernels (the torch kernels in the same process only generate the synthetic input)."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
keep = [r for r in rows if r["Name"].startswith("pjb::") or "pjb::" in r["Name"].split("(")[0]]
tot = sum(int(r["TotalDurationNs"]) for r in keep)
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "PctOfPjbKernels"])
    for r in sorted(keep, key=lambda r: -int(r["TotalDurationNs"])):
        name = r["Name"].split("(")[0].replace("void ", "")
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"],
                    f"{100.0 * int(r['TotalDurationNs']) / tot:.2f}"])
print(f"{len(keep)} pjb kernels, total {tot / 1e6:.3f} ms")
