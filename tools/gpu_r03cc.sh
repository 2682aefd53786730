#!/bin/bash
# the host reader's own DEFLATE decoder against zlib: bamfilt and junc --ingest host on the configs[1] files; tests that read files
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bamfilt.py tests/test_gpu_host_cli.py -x -q 2>&1 | tail -2
g++ -O2 -std=c++17 -Iportcullis_amd/host/include -o /tmp/fi_speed tests/cpp/fast_inflate_check.cc portcullis_amd/host/src/fast_inflate.cc -lz && /tmp/fi_speed speed 512
for k in 1 2; do
python tools/bench_bamfilt_program.py --runs 7 --env PORTCULLIS_ZLIB_INFLATE=1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bamfilt zlib       ', sorted(d['wall_s']), d['kept_bytes_md5'])"
python tools/bench_bamfilt_program.py --runs 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bamfilt fastInflate', sorted(d['wall_s']), d['kept_bytes_md5'])"
done
wd=/tmp/pjb_bamfilt
exe=portcullis_amd/host/portcullis_amd
python - <<'PY'
import subprocess, time, os, hashlib
wd='/tmp/pjb_bamfilt'; exe='portcullis_amd/host/portcullis_amd'
for label, env in (('junc --ingest host zlib', {'PORTCULLIS_ZLIB_INFLATE':'1'}), ('junc --ingest host fastInflate', {}))*2:
    ts=[]
    for k in range(5):
        t=time.time(); p=subprocess.run([exe,'junc','--ingest','host','-t','16','-o',wd+'/oh/pc',wd+'/prep'],capture_output=True,text=True,env=dict(os.environ,**env)); ts.append(time.time()-t)
        assert p.returncode==0, p.stderr[-500:]
    print(label, sorted(round(x,3) for x in ts), hashlib.md5(open(wd+'/oh/pc.junctions.tab','rb').read()).hexdigest())
PY
PORTCULLIS_PROFILE_PIECES=1 PORTCULLIS_PROFILE=1 $exe bamfilt -o $wd/filt/f2.bam -c HARD -t 16 $wd/pass.junctions.tab $wd/prep/portcullis.sorted.alignments.bam 2>&1 | grep "profile\|scan piece" | head -24
