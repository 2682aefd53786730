#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_cli.py tests/test_gpu_bamfilt.py -x -q 2>&1 | tail -4
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03ba_bench.json 2> gpurun_out/r03ba_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03ba_variants.txt
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
    print(f'{label:44s} median {sorted(ts)[len(ts)//2]:.3f}  runs {[round(t, 3) for t in ts]}', flush=True)
run('child does the work (default)', {})
run('one process', {'PORTCULLIS_NO_FORK': '1'})
run('child does the work (default)', {})
run('one process', {'PORTCULLIS_NO_FORK': '1'})
PY
python tools/bench_bamfilt_program.py --runs 5 | cut -c1-330
