#!/bin/bash
# usage (under gpurun): tools/profile_ingest.sh <tag>  -- rocprofv3 kernel stats of the device ingest on the C2 BAM
TAG=$1
cd $GRAFT_REPO_ROOT
PORTCULLIS_INGEST=device python tests/e2e_bench.py --config C2 --threads 16 --workdir /tmp/pi --keep --no-oracle --repeat 1 > /dev/null 2>&1
python3 tools/ingest_profile.py /tmp/pi/prep 3
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/tools/ingest_profile.py /tmp/pi/prep 3 > $OUT/ingest_prof_$TAG.log 2>&1
ls $OUT/prof_$TAG | head
rm -rf /tmp/pi
