#!/bin/bash
# junc end to end: piece size and ring depth once more
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03cn_bench.json 2> gpurun_out/r03cn_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03cn_pieces.txt
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
    for env in ({}, {'PORTCULLIS_PIECE_MB': '32'}, {'PORTCULLIS_PIECE_MB': '128'}, {'PORTCULLIS_PINNED_BUFFERS': '8'}, {'PORTCULLIS_PINNED_BUFFERS': '24'}, {'PORTCULLIS_PIECE_MB': '32', 'PORTCULLIS_PINNED_BUFFERS': '24'}):
        run('default' if not env else ' '.join(k[11:] + '=' + v for k, v in env.items()), env)
PY

