#!/bin/bash
# round 3, call j: end-to-end variants on the prepared 33 GB BAM, 5 runs each (wall clock by python)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
PJB_BENCH_E2E_REPS=1 PJB_BENCH_NO_E2E_CPU=1 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03j_bench.json 2> gpurun_out/r03j_bench.err
python - <<'PY' 2>&1 | tee gpurun_out/r03j_variants.txt
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
run('default (warm-up)', {}, 3)
run('default', {})
run('genome_early', {'PORTCULLIS_GENOME_EARLY': '1'})
run('piece32', {'PORTCULLIS_PIECE_MB': '32'})
run('piece32 x 20', {'PORTCULLIS_PIECE_MB': '32', 'PORTCULLIS_PINNED_BUFFERS': '20'})
run('piece16 x 24', {'PORTCULLIS_PIECE_MB': '16', 'PORTCULLIS_PINNED_BUFFERS': '24'})
run('buffers 8', {'PORTCULLIS_PINNED_BUFFERS': '8'})
run('slots 2', {'PORTCULLIS_TRANSFER_SLOTS': '2'})
run('slots 4', {'PORTCULLIS_TRANSFER_SLOTS': '4'})
run('inflate v1', {'PJB_INFLATE_V1': '1'})
run('default again', {})
PY
