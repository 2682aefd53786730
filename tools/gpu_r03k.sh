#!/bin/bash
# round 3, call k: target groups (one kernel chain over several targets): parity tests, then the bench line with and without
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -q -x 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r03k_pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --group-bases 0 > gpurun_out/r03k_bench_singles.json 2> gpurun_out/r03k_bench_singles.err ) 2>&1 | tail -3
tail -c 300 gpurun_out/r03k_bench_singles.err
( time timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r03k_bench_groups.json 2> gpurun_out/r03k_bench_groups.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/r03k_bench_groups.err
python - <<'PY'
import json
for tag in ('singles', 'groups'):
    try:
        d = json.load(open(f'gpurun_out/r03k_bench_{tag}.json'))
    except Exception as e:
        print(tag, 'no line', e); continue
    r = d['roofline']
    print(tag, 'value %.2f G reads/s' % (d['value'] / 1e9), 'ms/step %.2f' % d['ms_per_step'], 'step_frac', r['step_frac'], 'kernel ms/step', d['device_kernel_ms_per_step'], 'dominant', r['kernel'], r.get('frac'), r.get('frac_alone'))
    for k in d['kernels'][:14]: print('   ', k)
    if d.get('cpu_baseline'): print('  cpu', d['cpu_baseline']['value'])
    if d.get('e2e'): print('  e2e', d['e2e'].get('wall_s'), d['e2e'].get('runs_s'))
PY
python - <<'PY' 2>&1 | tee gpurun_out/r03k_variants.txt
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
    print(f'{label:28s} median {sorted(ts)[len(ts)//2]:.3f}  runs {[round(t, 3) for t in ts]}', flush=True)
run('default', {})
run('two contexts', {'PORTCULLIS_CTX_PER_GPU': '2'})
run('two contexts, slots 4', {'PORTCULLIS_CTX_PER_GPU': '2', 'PORTCULLIS_TRANSFER_SLOTS': '4'})
run('host queue 1', {'PJB_HOST_QUEUE': '1'})
PY
