#!/bin/bash
# round 3, call t: device thread never waits for a chain (pjb_finish_ready), deeper queue, slot released before BAMEND is queued
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_host_cli.py tests/test_gpu_parity.py -q -x 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r03t_pytest.log

PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03t_bench.json 2> gpurun_out/r03t_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03t_variants.txt
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
run('host queue 2', {'PJB_HOST_QUEUE': '2'})
run('inflate 4 wg/cu', {'PJB_INFLATE_WG_PER_CU': '4'})
run('inflate 3 wg/cu', {'PJB_INFLATE_WG_PER_CU': '3'})
run('default again', {})
PY
PJB_PROFILE_HOST=2 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o /tmp/pjb_bench_e2e/out/pc2 /tmp/pjb_bench_e2e/prep > /dev/null 2> gpurun_out/r03t_host.txt
grep -E "device thread|workers|main:|context ready" gpurun_out/r03t_host.txt
