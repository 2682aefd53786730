#!/bin/bash
# round 3, call r: hardware-queue count for the end-to-end program; full GPU suite at this commit
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r03r_pytest.log
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03r_bench.json 2> gpurun_out/r03r_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03r_variants.txt
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
run('default (8 queues)', {})
run('4 queues', {'GPU_MAX_HW_QUEUES': '4'})
run('16 queues', {'GPU_MAX_HW_QUEUES': '16'})
run('24 queues', {'GPU_MAX_HW_QUEUES': '24'})
run('8 queues again', {})
PY
PJB_PROFILE_HOST=2 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o /tmp/pjb_bench_e2e/out/pc2 /tmp/pjb_bench_e2e/prep > /dev/null 2> gpurun_out/r03r_host.txt
grep -E "device thread|workers|main:|context ready" gpurun_out/r03r_host.txt
