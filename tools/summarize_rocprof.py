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
if len(sys.argv) > 3:  # also as JSON (profiles/rocprof_latest.json: bench.py's frac_rocprof), with the hash of the kernel sources
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from csrc_hash import csrc_hash
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kernels = {}
    for r in keep:
        name = r["Name"].split("(")[0].replace("void ", "").split("<")[0]
        k = kernels.setdefault(name, dict(calls=0, total_ns=0))
        k["calls"] += int(r["Calls"])
        k["total_ns"] += int(r["TotalDurationNs"])
    for k in kernels.values():
        k["avg_ns"] = k["total_ns"] / max(k["calls"], 1)
    json.dump({"_workload": sys.argv[4] if len(sys.argv) > 4 else None, "_commit": sys.argv[5] if len(sys.argv) > 5 else None,
               "_source": os.path.basename(dst), "_csrc_hash": csrc_hash(root), "kernels": kernels}, open(sys.argv[3], "w"), indent=1)
