# (experiment, round 6) the program's slow state (4 - 6 s instead of 2): is the BAM still in the page cache?  Before every run: the cgroup's
# memory numbers and the fraction of the file that is resident (mincore on a mapping).
OUT=gpurun_out; mkdir -p $OUT
PJB_BENCH_E2E_REPS=1 PJB_BENCH_E2E_EARLY_REPS=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench e2e runs', d['e2e'].get('runs_s'), d['e2e'].get('error'))"
BAM=/tmp/pjb_bench_e2e/prep/portcullis.sorted.alignments.bam
for f in memory.max memory.high memory.current memory.swap.max; do echo "$f $(cat /sys/fs/cgroup/$f 2>/dev/null)"; done
free -g | head -3
res() { python3 - $BAM <<'PY'
import ctypes, mmap, os, sys
fd = os.open(sys.argv[1], os.O_RDONLY); n = os.fstat(fd).st_size
libc = ctypes.CDLL(None, use_errno=True)
libc.mmap.restype = ctypes.c_void_p
libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
p = libc.mmap(None, n, mmap.PROT_READ, mmap.MAP_SHARED, fd, 0)
pages = (n + 4095) // 4096
vec = (ctypes.c_ubyte * pages)()
libc.mincore.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
rc = libc.mincore(ctypes.c_void_p(p), n, vec)
r = sum(b & 1 for b in bytes(vec))
d = {k: int(v) for k, v in (l.split() for l in open('/sys/fs/cgroup/memory.stat'))}
print('resident %.1f %% of %.1f GB; cgroup: current %.1f GB, file %.1f (active %.1f, inactive %.1f), anon %.1f, unevictable %.1f' % (
    100.0 * r / pages, n / 1e9, int(open('/sys/fs/cgroup/memory.current').read()) / 1e9, d['file'] / 1e9, d['active_file'] / 1e9, d['inactive_file'] / 1e9, d['anon'] / 1e9, d['unevictable'] / 1e9))
PY
}
EXE=portcullis_amd/host/portcullis_amd
for k in 1 2 3 4 5 6 7 8 9 10; do
  res
  s=$(date +%s.%N); env ${ENVX:-PJB_NONE=1} $EXE junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/io /tmp/pjb_bench_e2e/prep > /dev/null 2>&1; e=$(date +%s.%N)
  python3 -c "print('e2e run $k: %.3f s' % ($e - $s))"; done 2>&1 | tee $OUT/r06_e2e_residency.txt
res | tee -a $OUT/r06_e2e_residency.txt
