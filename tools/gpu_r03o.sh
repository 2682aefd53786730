#!/bin/bash
# round 3, call o: configs[4] whole on one GPU (test + bench --config c5), hardware-queue count A/B on the step and the e2e run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "c5_whole" 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r03o_pytest_c5.log
( time timeout 1500 python bench.py --config c5 --steps 5 --warmup 2 > gpurun_out/r03o_bench_c5.json 2> gpurun_out/r03o_bench_c5.err ) 2>&1 | tail -3
tail -c 400 gpurun_out/r03o_bench_c5.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r03o_bench_c5.json'))
    print('c5 value',d['value'],'ms',d['ms_per_step'],'step_frac',d['roofline']['step_frac'], d['config']['workload'][:120], 'cpu', (d['cpu_baseline'] or {}).get('value'))
except Exception as e: print('c5 bench failed', e)
PY
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline > gpurun_out/r03o_bench_q$q.json 2> gpurun_out/r03o_bench.err || tail -c 300 gpurun_out/r03o_bench.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03o_bench_q$q.json'))
print('GPU_MAX_HW_QUEUES $q ms/step %.2f' % d['ms_per_step'], 'kernel ms/step %.2f' % d['device_kernel_ms_per_step'], 'step_frac', d['roofline']['step_frac'])
PY
done 2>&1 | tee gpurun_out/r03o_hwq.txt
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03o_bench.json 2> gpurun_out/r03o_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03o_variants.txt
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
run('hw queues 8', {'GPU_MAX_HW_QUEUES': '8'})
run('two contexts', {'PORTCULLIS_CTX_PER_GPU': '2'})
run('two contexts, hw queues 8', {'PORTCULLIS_CTX_PER_GPU': '2', 'GPU_MAX_HW_QUEUES': '8'})
run('default again', {})
PY
