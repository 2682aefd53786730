python tests/e2e_bench.py --config C2 --contigs 4 --threads 128 --keep --no-oracle --repeat 1 --workdir /tmp/pjb_e2e4 2>&1 | tail -1
for t in 32 128; do echo "== threads $t"; PJB_PROFILE_HOST=1 portcullis_amd/host/portcullis_amd junc -t $t -o /tmp/pjb_e2e4/o$t/pc /tmp/pjb_e2e4/prep 2>&1 | grep -E "host profile|Wall time"; done
