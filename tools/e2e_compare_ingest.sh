#!/bin/bash
# usage (under gpurun): tools/e2e_compare_ingest.sh [config] [contigs] [repeats]
# Wall clock of `portcullis_amd junc` on one synthetic prepared BAM with host ingest (decode threads) and
# device ingest (pjb_submit_bam); the two .tab files must be identical.
CFG=${1:-C2}; NC=${2:-1}; REP=${3:-3}
cd $GRAFT_REPO_ROOT
cat /sys/fs/cgroup/memory.max 2>/dev/null | sed 's/^/memory.max /'
PORTCULLIS_INGEST=host python tests/e2e_bench.py --config $CFG --contigs $NC --threads 16 --workdir /tmp/e2e --keep --no-oracle --repeat $REP > gpurun_out/e2e_host.json 2> gpurun_out/e2e_host.err || { tail -5 gpurun_out/e2e_host.err; exit 1; }
rm -rf /tmp/e2e/contig*   # the SoA dumps are no longer needed
ls -la /tmp/e2e/prep | tail -4
for i in $(seq $REP); do
  s=$(date +%s%N); PJB_PROFILE_HOST=1 PORTCULLIS_INGEST=device portcullis_amd/host/portcullis_amd junc -t 16 -o /tmp/e2e/outd/pc /tmp/e2e/prep > /tmp/e2e/d.log 2> /tmp/e2e/d.err; e=$(date +%s%N)
  echo "device ingest wall $(( (e - s) / 1000000 )) ms"; grep "host profile\] \(process\|main\|workers\)" /tmp/e2e/d.err
done
md5sum /tmp/e2e/outd/pc.junctions.tab /tmp/e2e/out$((REP-1))/pc.junctions.tab
echo "host ingest: $(cut -c1-330 gpurun_out/e2e_host.json)"
rm -rf /tmp/e2e
