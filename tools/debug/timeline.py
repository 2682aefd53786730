"""Timeline of the last bench step from a rocprofv3 kernel trace: where the time of a step goes (kernel by kernel,
with real start/end timestamps, overlap and idle gaps).  usage: timeline.py <kernel_trace.csv> [n_contigs]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].split()[-1]
        if name.endswith(".kd"):
            name = name[:-3]
        name = name.replace("pjb::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
# last step: the last n k1_count launches
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 25
starts = [i for i, r in enumerate(rows) if r[2] in ("k1_count", "k1_walk")]
if len(starts) < nc:
    print("kernel names seen:", sorted(set(r[2] for r in rows))[:40])
    sys.exit(1)
first = starts[-nc]
# a step starts at the first k1_count of the contig set; the one before it for comparison
seg = rows[first:]
t0, t1 = seg[0][0], max(r[1] for r in seg)
print(f"last step: {len(seg)} launches, {(t1 - t0) / 1e6:.3f} ms wall")
# union of busy time, idle
ev = sorted((r[0], r[1]) for r in seg)
busy = 0
cur_s, cur_e = ev[0]
gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"busy (union) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
gs = sorted(g[0] for g in gaps)
if gs:
    print("gap ns: median %d, p90 %d, max %d; gaps > 20 us: %d (sum %.3f ms)" % (gs[len(gs) // 2], gs[int(len(gs) * .9)], gs[-1],
          sum(1 for g in gs if g > 20000), sum(g for g in gs if g > 20000) / 1e6))
per = defaultdict(lambda: [0, 0])
for s, e, n, q in seg:
    per[n][0] += 1
    per[n][1] += e - s
print("kernel                 launches  total_ms  avg_us")
for n, (k, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:24s} {k:6d} {t / 1e6:9.3f} {t / k / 1e3:8.1f}")
print("sum of kernel durations %.3f ms" % (sum(v[1] for v in per.values()) / 1e6))
# gap attributed to the kernel that follows it
after = defaultdict(lambda: [0, 0])
prev_end = seg[0][0]
for s, e, n, q in sorted(seg):
    if s > prev_end:
        after[n][0] += 1
        after[n][1] += s - prev_end
    prev_end = max(prev_end, e)
print("idle before kernel     count  total_ms  avg_us")
for n, (k, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"{n:24s} {k:6d} {t / 1e6:9.3f} {t / k / 1e3:8.1f}")
import os
if os.environ.get("TIMELINE_LIST"):  # every launch of the step in start order: start, end, duration (us from the step's start), stream
    print("start_us   end_us   dur_us  kernel (stream)")
    for s, e, n, q in sorted(seg):
        print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  {n} ({q})")
