#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; values in KiB).
gfx950 correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 64 B per 128-B request for
wide coalesced streams, i.e. reports 1/2 of the bytes -> doubled here; WRITE_SIZE is taken as is."""
import csv
import json
import sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else None
commit = sys.argv[4] if len(sys.argv) > 4 else None
res = defaultdict(lambda: defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{d}/{ctr}/pmc_counter_collection.csv")):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]  # template arguments dropped
        if "pjb::" not in name or r["Counter_Name"] != ctr:
            continue
        res[name][ctr].append(float(r["Counter_Value"]))
summary = {}
for name, v in sorted(res.items()):
    f = v.get("FETCH_SIZE", [])
    w = v.get("WRITE_SIZE", [])
    fk = sum(f) / len(f) if f else 0.0
    wk = sum(w) / len(w) if w else 0.0
    summary[name] = dict(launches=len(f), fetch_kib_raw=round(fk, 1), write_kib=round(wk, 1),
                         hbm_bytes_per_launch=int((2 * fk + wk) * 1024))
meta = {}
if workload:
    meta["_workload"] = workload  # bench.py attaches the traffic figure only to a run of the same workload
if commit:
    meta["_commit"] = commit
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_hash
meta["_csrc_hash"] = csrc_hash(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
json.dump({**meta, **summary}, open(out, "w"), indent=1)
for k, v in summary.items():
    print(f"{k:45s} fetch_raw {v['fetch_kib_raw'] / 1024:9.1f} MiB  write {v['write_kib'] / 1024:9.1f} MiB  corrected {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB")
