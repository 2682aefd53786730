#!/bin/bash
# round 3, call n: whole GPU suite, long-read datapoint, then rocprofv3 stats + PMC traffic + the bench line at this commit
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r03cg_pytest.log
( time timeout 600 python tools/bench_long_reads.py > gpurun_out/r03cg_long_reads.json 2> gpurun_out/r03cg_long_reads.err ) 2>&1 | tail -3
tail -c 300 gpurun_out/r03cg_long_reads.err; cat gpurun_out/r03cg_long_reads.json
bash tools/profile_round.sh r03cg $1 2>&1 | tail -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03cg_bench_C3.json'))
print('value',d['value'],'ms',d['ms_per_step'],'step_frac',d['roofline']['step_frac'], d['roofline']['kernel'], d['roofline'].get('frac'), d['roofline'].get('frac_alone'), d['roofline'].get('traffic'))
print('cpu', d['cpu_baseline']['value'], 'e2e', json.dumps(d['e2e'])[:900])
PY
python tools/bench_bamfilt_program.py --runs 9 > gpurun_out/r03cg_bamfilt_program.json 2> gpurun_out/r03cg_bamfilt_program.err
cut -c1-400 gpurun_out/r03cg_bamfilt_program.json
python tools/bench_extra.py > gpurun_out/r03cg_extra.json 2>/dev/null; cut -c1-300 gpurun_out/r03cg_extra.json
