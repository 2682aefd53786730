"""Start / end of every bgzf_inflate launch in a rocprofv3 kernel trace (ms since the first GPU activity), with the
number of inflate kernels running at the start of each and the idle time before it."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
inf = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)) for r in rows if any(k in r["Kernel_Name"] for k in ("bgzf_inflate", "bgzf_decode", "bgzf_resolve")))
last_end = None
for a, b, q, g in inf:
    running = sum(1 for x, y, _, _ in inf if x < a < y)
    gap = (a - last_end) / 1e6 if last_end is not None else 0.0
    print(f"{(a - t0) / 1e6:8.1f} -> {(b - t0) / 1e6:8.1f}  {(b - a) / 1e6:6.1f} ms  queue {q}  grid {g}  others running {running}  gap since last end {gap:7.1f} ms")
    last_end = max(last_end or 0, b)
