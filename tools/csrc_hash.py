#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the kernel sources (portcullis_amd/csrc/*.hip, *.hip.h, Makefile; comments included):
bench.py quotes a committed PMC / rocprof figure only if it was measured on exactly these files."""
import glob
import hashlib
import os
import sys


def csrc_hash(root):
    h = hashlib.sha256()
    d = os.path.join(root, "portcullis_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hip.h")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash(sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
