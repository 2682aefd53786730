#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 -o /tmp/ring_probe tools/debug/ring_probe.cc -lpthread 2>&1 | tail -2
{
/tmp/ring_probe /tmp/rp.bin 12 3 5 0 64 > /dev/null; sync
for rep in 1 2; do
echo "== reads only, 3 groups x 5 threads"; /tmp/ring_probe /tmp/rp.bin 12 3 5 0 64
echo "== + DMA"; /tmp/ring_probe /tmp/rp.bin 12 3 5 1 64
echo "== + DMA, GPU streaming through HBM meanwhile"; /tmp/ring_probe /tmp/rp.bin 12 3 5 1 64 0 0 2
echo "== + DMA, 1 group x 15 threads"; /tmp/ring_probe /tmp/rp.bin 12 1 15 1 64
echo "== + DMA, 5 groups x 3 threads"; /tmp/ring_probe /tmp/rp.bin 12 5 3 1 64
echo "== + DMA, 3 groups x 5 threads, 16 MB pieces"; /tmp/ring_probe /tmp/rp.bin 12 3 5 1 16
done
} 2>&1 | tee gpurun_out/r03q_ring_probe.txt
