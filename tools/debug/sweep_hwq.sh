# (experiment) hardware queues the runtime maps the streams onto
for q in 4 2 6 8 4; do echo "== GPU_MAX_HW_QUEUES $q"; GPU_MAX_HW_QUEUES=$q python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline > gpurun_out/r05az_q$q.json 2>/dev/null; python3 tools/show_bench.py gpurun_out/r05az_q$q.json | head -4 | grep -E "ms_per_step|overlap"; done
