#!/bin/bash
# round 3, call i: end-to-end variants on the prepared 33 GB BAM (after the bench line's own 5 runs)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python bench.py --steps 10 --warmup 3 > gpurun_out/r03i_bench.json 2> gpurun_out/r03i_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03i_bench.json'))
e=d['e2e']; print('e2e median', e['wall_s'], e['runs_s'], 'cpu', e.get('cpu',{}).get('wall_s'), 'md5 ok', e['tab_identical_to_oracle'])
PY
W=/tmp/pjb_bench_e2e
REF=$(md5sum $W/out/pc.junctions.tab | cut -d' ' -f1)
run() { # label, env...
  local label=$1; shift
  for k in 1 2 3; do
    local t0=$(date +%s.%N)
    env "$@" portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/v $W/prep > /dev/null 2> /tmp/v.err
    local t1=$(date +%s.%N)
    local m=$(md5sum $W/out/v.junctions.tab | cut -d' ' -f1)
    echo "$label run $k: $(echo "$t1 - $t0" | bc) s  md5 $([ "$m" = "$REF" ] && echo same || echo DIFFERENT)"
  done
}
{
run default X=1
run genome_early PORTCULLIS_GENOME_EARLY=1
run piece32 PORTCULLIS_PIECE_MB=32
run piece32x16 PORTCULLIS_PIECE_MB=32 PORTCULLIS_PINNED_BUFFERS=16
run slots2 PORTCULLIS_TRANSFER_SLOTS=2
run slots4 PORTCULLIS_TRANSFER_SLOTS=4
run inflate_v1 PJB_INFLATE_V1=1
} 2>&1 | tee gpurun_out/r03i_variants.txt
PJB_PROFILE_HOST=2 portcullis_amd/host/portcullis_amd junc -t 16 --orientation FR -o $W/out/pc2 $W/prep > /dev/null 2> gpurun_out/r03i_host.txt
grep -E "device thread|workers|main:|context ready" gpurun_out/r03i_host.txt
