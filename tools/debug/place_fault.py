#!/usr/bin/env python3
"""place_fault.py <address> <log>: the device buffers ([alloc] lines of a PJB_DEBUG_ALLOC build) against a faulting address: the
nearest ones, and for every buffer the element index the address would be for 4 / 8 / 32 / 192-byte elements (a wild index
usually looks like something: 0xffffxxxx, a pair count, ...)."""
import re
import sys

addr = int(sys.argv[1], 16)
rows = {}
for line in open(sys.argv[2], errors="replace"):
    m = re.search(r"\[alloc\] (.*?): (0x[0-9a-f]+) \.\. (0x[0-9a-f]+)", line)
    if m:
        rows[m.group(1)] = (int(m.group(2), 16), int(m.group(3), 16))
for name, (lo, hi) in sorted(rows.items(), key=lambda kv: kv[1]):
    d = addr - lo
    if d < 0:
        continue
    idx = {e: d // e for e in (4, 8, 32, 192)}
    print(f"  {name:40s} [{lo:#x}, +{hi - lo:#x})  +{d:#x}: " + "  ".join(f"/{e}={v:#x}" for e, v in idx.items()))
