#!/bin/bash
# junc end to end: targets read their first pieces while the device context comes up (A/B on one box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_cli.py -x -q 2>&1 | tail -3
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03bx_bench.json 2> gpurun_out/r03bx_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03bx_variants.txt
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
run('early read (default)', {})
run('wait for the context first', {'PORTCULLIS_EARLY_READ': '0'})
run('early read (default)', {})
run('wait for the context first', {'PORTCULLIS_EARLY_READ': '0'})
run('early read, 12 GB', {'PORTCULLIS_EARLY_MB': '12288'})
run('early read, 3 GB', {'PORTCULLIS_EARLY_MB': '3072'})
p = subprocess.run([cli, 'junc', '-t', '16', '--orientation', 'FR', '-o', W + '/out/v', W + '/prep'], capture_output=True, text=True, env=dict(os.environ, PJB_PROFILE_HOST='1'))
open('gpurun_out/r03bx_host_profile.txt', 'w').write(p.stdout + p.stderr)
PY
grep "t=\|device thread:" gpurun_out/r03bx_host_profile.txt | head; grep -c "early read" gpurun_out/r03bx_host_profile.txt
