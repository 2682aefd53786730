#!/bin/bash
# first GPU call of the round: disk/cores, GPU tests, the bench line with every leg, the N=2 code path on one GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
df -h /tmp | tail -2; nproc; cat /sys/fs/cgroup/cpu.max; free -g | head -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
( time python bench.py --steps 20 --warmup 3 > gpurun_out/r02a_bench_C3.json 2> gpurun_out/r02a_bench.err ) 2>&1 | tail -4
tail -c 600 gpurun_out/r02a_bench.err
cat gpurun_out/r02a_bench_C3.json | cut -c1-3000
( time PJB_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --reads 20000000 --junctions 25000 > gpurun_out/r02a_bench_share2.json 2> gpurun_out/r02a_share2.err ) 2>&1 | tail -4
tail -c 600 gpurun_out/r02a_share2.err
cut -c1-1500 gpurun_out/r02a_bench_share2.json
