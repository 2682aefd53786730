python tools/e2e_bench.py --config C2 --threads ${1:-64} --keep --no-oracle --repeat 1 --workdir /tmp/pjb_e2e 2>&1 | tail -1
for t in 8 32 64 128; do echo "== threads $t"; /usr/bin/env PJB_PROFILE_HOST=1 portcullis_amd/host/portcullis_amd junc -t $t -o /tmp/pjb_e2e/o$t/pc /tmp/pjb_e2e/prep 2>&1 | grep -E "host profile|Wall time" ; done
