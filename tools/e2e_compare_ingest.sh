#!/bin/bash
# usage (under gpurun): tools/e2e_compare_ingest.sh [config] [contigs]  -- wall clock of `junc` with host and device ingest
CFG=${1:-C2}; NC=${2:-1}
cd $GRAFT_REPO_ROOT
python tools/e2e_bench.py --config $CFG --contigs $NC --threads 16 --workdir /tmp/e2e --keep --no-oracle --repeat 3 > gpurun_out/e2e_host.json 2> gpurun_out/e2e_host.err
for i in 1 2 3; do
  s=$(date +%s%N); PJB_PROFILE_HOST=1 PORTCULLIS_INGEST=device portcullis_amd/host/portcullis_amd junc -t 16 -o /tmp/e2e/outd/pc /tmp/e2e/prep > /tmp/e2e/d.log 2> /tmp/e2e/d.err; e=$(date +%s%N)
  echo "device ingest wall $(( (e - s) / 1000000 )) ms"; grep "host profile" /tmp/e2e/d.err | head -8
done
md5sum /tmp/e2e/outd/pc.junctions.tab /tmp/e2e/out2/pc.junctions.tab
cut -c1-330 gpurun_out/e2e_host.json
