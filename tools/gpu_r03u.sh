#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03u_bench.json 2> gpurun_out/r03u_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03u_variants.txt
import hashlib, os, subprocess, time
W = '/tmp/pjb_bench_e2e'
ref = hashlib.md5(open(W + '/out/pc.junctions.tab', 'rb').read()).hexdigest()
cli = 'portcullis_amd/host/portcullis_amd'
def run(label, env, n=7):
    ts = []
    for k in range(n):
        t = time.time()
        p = subprocess.run([cli, 'junc', '-t', '16', '--orientation', 'FR', '-o', W + '/out/v', W + '/prep'], capture_output=True, text=True, env=dict(os.environ, **env))
        ts.append(time.time() - t)
        same = hashlib.md5(open(W + '/out/v.junctions.tab', 'rb').read()).hexdigest() == ref
        if p.returncode or not same:
            print(label, 'FAILED', p.returncode, same, p.stderr[-300:])
    print(f'{label:28s} median {sorted(ts)[len(ts)//2]:.3f}  runs {[round(t, 3) for t in ts]}', flush=True)
run('default', {})
run('inflate 3 wg/cu', {'PJB_INFLATE_WG_PER_CU': '3'})
run('inflate 2 wg/cu', {'PJB_INFLATE_WG_PER_CU': '2'})
run('inflate 3 wg/cu, 4 hw queues', {'PJB_INFLATE_WG_PER_CU': '3', 'GPU_MAX_HW_QUEUES': '4'})
run('inflate 3, normal priority', {'PJB_INFLATE_WG_PER_CU': '3', 'PJB_INFLATE_NORMAL_PRIORITY': '1'})
PY
for w in 5 3; do
PJB_INFLATE_WG_PER_CU=$w PJB_PROFILE_HOST=2 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o /tmp/pjb_bench_e2e/out/pc2 /tmp/pjb_bench_e2e/prep > /dev/null 2> gpurun_out/r03u_host_wg$w.txt
grep -E "device thread|workers|main:|context ready" gpurun_out/r03u_host_wg$w.txt
done
