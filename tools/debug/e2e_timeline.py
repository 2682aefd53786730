"""Timeline of an end-to-end `portcullis_amd junc` run from rocprofv3's kernel and memory-copy traces: per 100 ms bin
the time some kernel was running (union), the time some copy was running (union), bytes copied, and the busiest kernels."""
import csv, glob, sys, collections
d = sys.argv[1]
def load(pat):
    rows = []
    for f in glob.glob(pat, recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows
K = load(f"{d}/**/*kernel_trace.csv")
M = load(f"{d}/**/*memory_copy_trace.csv")
if M: print("copy columns:", list(M[0].keys()))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in K]
ms = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), int(r.get("Size", r.get("Bytes", 0)) or 0)) for r in M]
t0 = min([a for a, _, _ in ks] + [a for a, _, _, _ in ms])
t1 = max([b for _, b, _ in ks] + [b for _, b, _, _ in ms])
print(f"span {(t1 - t0) / 1e9:.3f} s, {len(ks)} kernels, {len(ms)} copies, {sum(m[3] for m in ms) / 1e9:.2f} GB copied")
def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for a, b in iv:
        if cs is None: cs, ce = a, b
        elif a <= ce: ce = max(ce, b)
        else:
            tot += ce - cs; cs, ce = a, b
    if cs is not None: tot += ce - cs
    return tot
print(f"kernel busy (union) {union([(a, b) for a, b, _ in ks]) / 1e9:.3f} s; copy busy (union) {union([(a, b) for a, b, _, _ in ms]) / 1e9:.3f} s")
bydir = collections.defaultdict(lambda: [0, 0, []])
for a, b, dr, sz in ms:
    bydir[dr][0] += sz; bydir[dr][1] += 1; bydir[dr][2].append((a, b))
for dr, (sz, n, iv) in bydir.items():
    u = union(iv)
    print(f"  copies {dr}: {n} x, {sz / 1e9:.2f} GB, busy {u / 1e9:.3f} s -> {sz / max(u, 1):.1f} GB/s while busy")
tot = collections.defaultdict(lambda: [0, 0])
for a, b, n in ks:
    tot[n][0] += b - a; tot[n][1] += 1
for n, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {n:28s} {c:6d} x  {t / 1e6:9.1f} ms  avg {t / c / 1e3:9.1f} us")
B = 100_000_000
nb = (t1 - t0) // B + 1
print("bin(100ms) kernel_busy copy_busy GB_copied  top kernels")
for i in range(nb):
    lo, hi = t0 + i * B, t0 + (i + 1) * B
    kk = [(max(a, lo), min(b, hi)) for a, b, _ in ks if a < hi and b > lo]
    mm = [(max(a, lo), min(b, hi)) for a, b, _, _ in ms if a < hi and b > lo]
    gb = sum(sz * (min(b, hi) - max(a, lo)) / max(b - a, 1) for a, b, _, sz in ms if a < hi and b > lo) / 1e9
    per = collections.defaultdict(int)
    for a, b, n in ks:
        if a < hi and b > lo: per[n] += min(b, hi) - max(a, lo)
    top = " ".join(f"{n}={t / 1e6:.0f}" for n, t in sorted(per.items(), key=lambda kv: -kv[1])[:3])
    print(f"{i:3d} {union(kk) / 1e6:7.1f} {union(mm) / 1e6:7.1f} {gb:6.2f}  {top}")
