#!/usr/bin/env python3
"""Condensed view of a bench.py JSON line: step time, roofline block, per-kernel table (ms per step, alone / in-step)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.4g} {d['unit']}  junctions/s {d.get('junctions_per_sec', 0):.4g}")
r = d.get("roofline", {})
print("roofline:", {k: r.get(k) for k in ("kernel", "frac", "frac_alone", "step_alg_bytes", "step_frac", "step_alg_bytes_survey", "step_frac_survey")})
print("device_kernel_ms_per_step", d.get("device_kernel_ms_per_step"), "overlap", d.get("overlap_factor"), "launches", d.get("launches_per_step"))
for k in d.get("kernels", [])[:40]:
    print(f"  {k['name']:28s} n/step {k.get('launches_per_step', 0):6.1f}  ms/step {k.get('ms_per_step', 0):7.3f}  avg_us {1e3 * k.get('avg_ms', 0):8.1f}  GB/s {k.get('gbps') or 0:8.1f}")
if "e2e" in d and d["e2e"]:
    print("e2e:", {k: v for k, v in d["e2e"].items() if not isinstance(v, (dict, list))})
