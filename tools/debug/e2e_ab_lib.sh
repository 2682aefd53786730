# (experiment) the program on the bench's prepared BAM with two builds of the kernel library, alternating: is a change of the kernels a
# change of the end-to-end time?  (behind a `bench` run in the same call: /tmp/pjb_bench_e2e)
P=$GRAFT_REPO_ROOT/portcullis_amd/host/portcullis_amd
mkdir -p /tmp/pjb_bench_e2e/ab
TIMEFORMAT=%R
for k in 1 2 3 4 5 6; do
  for v in new old; do
    if [ $v = old ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/variants/old; else unset LD_LIBRARY_PATH; fi
    t=$( { time $P junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/ab/pc /tmp/pjb_bench_e2e/prep > /dev/null 2>&1 ; } 2>&1 )
    echo "$v $t $(md5sum < /tmp/pjb_bench_e2e/ab/pc.junctions.tab | cut -c1-8)"
  done
done
