#!/bin/bash
# junc end to end over the number of hardware queues (the program asks for 8)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03cq_bench.json 2> gpurun_out/r03cq_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03cq_hw_queues.txt
import hashlib, os, subprocess, time
W = '/tmp/pjb_bench_e2e'
ref = hashlib.md5(open(W + '/out/pc.junctions.tab', 'rb').read()).hexdigest()
cli = 'portcullis_amd/host/portcullis_amd'
def run(label, env, n=5):
    ts = []
    for k in range(n):
        t = time.time()
        p = subprocess.run([cli, 'junc', '-t', '16', '--orientation', 'FR', '-o', W + '/out/v', W + '/prep'], capture_output=True, text=True, env=dict(os.environ, **env))
        ts.append(time.time() - t)
        same = hashlib.md5(open(W + '/out/v.junctions.tab', 'rb').read()).hexdigest() == ref
        if p.returncode or not same:
            print(label, 'FAILED', p.returncode, same, p.stderr[-300:])
    print(f'{label:44s} median {sorted(ts)[len(ts)//2]:.3f}  runs {[round(t, 3) for t in sorted(ts)]}', flush=True)
for rep in range(2):
    for q in ('8', '4', '6', '12'):
        run('GPU_MAX_HW_QUEUES=' + q, {'GPU_MAX_HW_QUEUES': q})
PY

