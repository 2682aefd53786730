#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03v_bench.json 2> gpurun_out/r03v_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03v_variants.txt
import hashlib, os, subprocess, time
W = '/tmp/pjb_bench_e2e'
ref = hashlib.md5(open(W + '/out/pc.junctions.tab', 'rb').read()).hexdigest()
cli = 'portcullis_amd/host/portcullis_amd'
def run(label, env, n=6):
    ts = []
    for k in range(n):
        t = time.time()
        p = subprocess.run([cli, 'junc', '-t', '16', '--orientation', 'FR', '-o', W + '/out/v', W + '/prep'], capture_output=True, text=True, env=dict(os.environ, **env))
        ts.append(time.time() - t)
        same = hashlib.md5(open(W + '/out/v.junctions.tab', 'rb').read()).hexdigest() == ref
        if p.returncode or not same:
            print(label, 'FAILED', p.returncode, same, p.stderr[-300:])
    print(f'{label:36s} median {sorted(ts)[len(ts)//2]:.3f}  runs {[round(t, 3) for t in ts]}', flush=True)
OLD = {'PORTCULLIS_BLOCKING_COLLECT': '1', 'PORTCULLIS_LEAVE_AFTER_PUSH': '1', 'PORTCULLIS_CMD_CAP': '6', 'PORTCULLIS_WORKERS_AS_THREADS': '1'}
def without(*ks): return {k: v for k, v in OLD.items() if k not in ks}
run('all old', OLD)
run('all new', {})
run('old + poll collect', without('PORTCULLIS_BLOCKING_COLLECT'))
run('old + leave first', without('PORTCULLIS_LEAVE_AFTER_PUSH'))
run('old + cap 32', without('PORTCULLIS_CMD_CAP'))
run('old + worker per target', without('PORTCULLIS_WORKERS_AS_THREADS'))
run('new, workers as threads', {'PORTCULLIS_WORKERS_AS_THREADS': '1'})
run('all old again', OLD)
PY
