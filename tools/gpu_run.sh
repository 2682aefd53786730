#!/bin/bash
# One parametrised GPU-box script (run as: gpurun -- 'bash tools/gpu_run.sh <step> [<step> ...]').  Steps:
#   tests            the whole GPU suite (pytest -m gpu -x)
#   tests:<expr>     pytest -m gpu -k <expr>
#   smoke            __graft_entry__.smoke()
#   quick            bench.py without the BAM leg and the CPU baseline (10 steps)            -> gpurun_out/<TAG>_quick.json
#   quick:<variant>  the same with tools/variants/libpjb_<variant>.so (tools/build_variants.sh)
#   bench            bench.py with default flags, as the driver runs it                       -> gpurun_out/<TAG>_bench_C3.json
#   prof             tools/profile_round.sh: rocprofv3 kernel stats + FETCH/WRITE PMC passes + the bench line
#   sq               SQ issue / stall counters per kernel (tools/pmc_sq.sh), condensed by tools/show_sq.py
#   sqv:<variant>    instruction counts per kernel (the first two counter groups of pmc_sq.sh) of tools/variants/libpjb_<variant>.so
#   e2eprof          (behind `bench`) the program on the bench's prepared BAM with PJB_PROFILE_HOST=1, three runs -> gpurun_out/<TAG>_e2e_host_profile_k.txt
#   fuzz             the three fuzz campaigns (tests/fuzz_campaign.py, fuzz_groups.py, fuzz_extra.py)
#   inflate          tools/bench_inflate.py at ~6 k and at ~130 k blocks a launch                              -> gpurun_out/<TAG>_bench_inflate.json
#   rankshare[:N,N]  tools/rank_share.py: every rank's share of configs[2] on this one GPU (PJB_BENCH_AS_RANK)  -> gpurun_out/<TAG>_rank_share.json
#   cmd:<shell>      anything else
# TAG (environment, default r05) names the outputs; COMMIT is recorded in the PMC summary.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r06}
OUT=gpurun_out
mkdir -p $OUT
rc=0
for step in "$@"; do
  echo "=== $step"
  case "$step" in
    tests) ( time timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_full.log 2>&1 ); grep -v "^  File \"/usr/local/lib" $OUT/${TAG}_pytest_full.log | tail -60 | tee $OUT/${TAG}_pytest.log ;;
    tests:*) ( time timeout 2400 python -m pytest tests -m gpu -q -x -k "${step#tests:}" 2>&1 | tail -40 ) 2>&1 | tee $OUT/${TAG}_pytest_k.log ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ;;
    quick) timeout 900 python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline > $OUT/${TAG}_quick.json 2> $OUT/${TAG}_quick.err; tail -c 400 $OUT/${TAG}_quick.err
           python3 tools/show_bench.py $OUT/${TAG}_quick.json ;;
    quick:*) v=${step#quick:}; PJB_BENCH_ABLATION=1 PJB_LIB_PATH=$PWD/tools/variants/libpjb_$v.so timeout 900 python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline > $OUT/${TAG}_quick_$v.json 2> $OUT/${TAG}_quick_$v.err; tail -c 300 $OUT/${TAG}_quick_$v.err
           python3 tools/show_bench.py $OUT/${TAG}_quick_$v.json | head -12 ;;
    bench) ( time python bench.py > $OUT/${TAG}_bench_C3.json 2> $OUT/${TAG}_bench.err ) 2>&1 | tail -4; tail -c 400 $OUT/${TAG}_bench.err
           python3 tools/show_bench.py $OUT/${TAG}_bench_C3.json ;;
    prof) bash tools/profile_round.sh $TAG ${COMMIT:-unknown}; python3 tools/show_bench.py $OUT/${TAG}_bench_C3.json ;;
    sq) bash tools/pmc_sq.sh $TAG > $OUT/${TAG}_sq_summary.csv 2>&1; python3 tools/show_sq.py $OUT/sq_$TAG/summary.csv | tee $OUT/${TAG}_sq.txt ;;
    sqv:*) v=${step#sqv:}; PJB_BENCH_ABLATION=1 PJB_LIB_PATH=$PWD/tools/variants/libpjb_$v.so SQ_GROUPS=2 bash tools/pmc_sq.sh ${TAG}_$v > $OUT/${TAG}_sq_${v}_summary.csv 2>&1; python3 tools/show_sq.py $OUT/sq_${TAG}_$v/summary.csv | head -4 | tee $OUT/${TAG}_sq_$v.txt ;;
    e2eprof) # after a `bench` step in the same call (the prepared directory is kept in /tmp/pjb_bench_e2e): the program's own host profile, three runs
           for k in 1 2 3; do ( time PJB_PROFILE_HOST=1 PJB_CREATE_TRACE=1 portcullis_amd/host/portcullis_amd junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/pc /tmp/pjb_bench_e2e/prep ) > $OUT/${TAG}_e2e_host_profile_$k.txt 2>&1; grep -E "real|Wall" $OUT/${TAG}_e2e_host_profile_$k.txt | head -3; done
           grep -E "host profile" $OUT/${TAG}_e2e_host_profile_2.txt | grep -v "submit_bam\|\] chr" | tail -40
           # the two chain plans side by side (wall of the command, alternating): groups (the default for this input) / one chain per target
           for k in 1 2 3 4 5 6; do for plan in groups:0 targets:0 groups:536870912 groups:268435456; do p=${plan%%:*}; gb=${plan##*:}; s=$(date +%s.%N); PORTCULLIS_CHAIN_PLAN=$p PORTCULLIS_GROUP_BASES=$gb portcullis_amd/host/portcullis_amd junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/pc_$p /tmp/pjb_bench_e2e/prep > /dev/null 2>&1; e=$(date +%s.%N); python3 -c "print('e2e plan $plan: %.3f s' % ($e - $s))"; done; done | tee $OUT/${TAG}_e2e_plans.txt
           python3 - $OUT/${TAG}_e2e_plans.txt <<'PY' | tee -a $OUT/${TAG}_e2e_plans.txt
import sys, collections, statistics
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    if ln.startswith("e2e plan "):
        k, v = ln[9:].split(": ")
        d[k].append(float(v.split()[0]))
for k, v in d.items():
    print(f"median {k}: {statistics.median(v):.3f} s  (min {min(v):.3f}, max {max(v):.3f}, {len(v)} runs)")
PY
           md5sum /tmp/pjb_bench_e2e/prof/pc_groups.junctions.tab /tmp/pjb_bench_e2e/prof/pc_targets.junctions.tab | tee -a $OUT/${TAG}_e2e_plans.txt ;;
    fuzz) ( timeout 1500 python tests/fuzz_campaign.py --seeds ${FUZZ_SEEDS:-100} --start ${FUZZ_START:-1000}; timeout 900 python tests/fuzz_groups.py --seeds ${FUZZ_GROUP_SEEDS:-100}; timeout 900 python tests/fuzz_extra.py ) 2>&1 | tail -30 | tee $OUT/${TAG}_fuzz.txt ;;
    fuzzbig) # the round's campaign: FUZZ_SEEDS seeds of fuzz_campaign.py in four processes side by side, then the group and the --extra campaigns
           n=${FUZZ_SEEDS:-1200}; q=$((n / 4))
           for k in 0 1 2 3; do ( timeout 3000 python tests/fuzz_campaign.py --seeds $q --start $((${FUZZ_START:-20000} + k * q)) 2>&1 | tail -8 > $OUT/${TAG}_fuzz_part$k.txt ) & done; wait
           ( cat $OUT/${TAG}_fuzz_part?.txt; timeout 1500 python tests/fuzz_groups.py --seeds ${FUZZ_GROUP_SEEDS:-300} --start 9000; timeout 1500 python tests/fuzz_extra.py ) 2>&1 | tail -40 | tee $OUT/${TAG}_fuzz.txt ;;
    e2eab) # (behind `bench`) round 5's program (tools/variants/r05, built from commit 4d50d07 by hand) against this tree's, alternating: wall of the command
           for k in 1 2 3 4 5 6; do for who in r05 now now_targets; do
             exe=portcullis_amd/host/portcullis_amd; plan=groups
             [ $who = r05 ] && exe=tools/variants/r05/host/portcullis_amd
             [ $who = now_targets ] && plan=targets
             s=$(date +%s.%N); PORTCULLIS_CHAIN_PLAN=$plan $exe junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/ab_$who /tmp/pjb_bench_e2e/prep > /dev/null 2>&1; e=$(date +%s.%N)
             python3 -c "print('e2e $who: %.3f s' % ($e - $s))"; done; done | tee $OUT/${TAG}_e2e_ab.txt
           python3 - $OUT/${TAG}_e2e_ab.txt <<'PY' | tee -a $OUT/${TAG}_e2e_ab.txt
import sys, collections, statistics
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    if ln.startswith("e2e "):
        k, v = ln[4:].split(": ")
        d[k].append(float(v.split()[0]))
for k, v in d.items():
    print(f"median {k}: {statistics.median(v):.3f} s  (min {min(v):.3f}, max {max(v):.3f}, {len(v)} runs)")
PY
           md5sum /tmp/pjb_bench_e2e/prof/ab_*.junctions.tab | tee -a $OUT/${TAG}_e2e_ab.txt ;;
    inflate) # bgzf_decode at both launch sizes: ~6 k blocks a launch (the C2 file in 256 MB chunks) and ~130 k (the file four times over, one launch)
           python tools/bench_inflate.py > $OUT/${TAG}_inflate_small.json 2> $OUT/${TAG}_inflate.err
           python tools/bench_inflate.py --chunk-mb 8192 --times 4 > $OUT/${TAG}_inflate_large.json 2>> $OUT/${TAG}_inflate.err
           python3 - "$OUT/${TAG}_inflate_small.json" "$OUT/${TAG}_inflate_large.json" "$OUT/${TAG}_bench_inflate.json" <<'PY'
import json, sys
small, large = (json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:3])
out = {"what": "device-side BGZF inflate (pjb_inflate_bgzf: bgzf_decode, a lane per block, + bgzf_resolve) on the configs[1] BAM; kernel time by HIP events; "
               "the rate depends on the blocks per launch: a launch needs >= 64 k blocks to fill the chip's lanes",
       "small_launches": {k: small[k] for k in ("blocks", "kernel_launches", "kernel_ms", "kernel_gbps_inflated", "kernel_gbps_compressed", "kernels")},
       "large_launch": {k: large[k] for k in ("blocks", "kernel_launches", "kernel_ms", "kernel_gbps_inflated", "kernel_gbps_compressed", "kernels")},
       "zlib_1thread_gbps_inflated": small["zlib_1thread_gbps_inflated"], "e2e": small.get("e2e")}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: {"blocks_per_launch": v["blocks"] // max(v["kernel_launches"], 1), "gbps_inflated": v["kernel_gbps_inflated"]} for k, v in out.items() if isinstance(v, dict) and "blocks" in v}))
PY
           ;;
    rankshare) timeout 3000 python tools/rank_share.py $TAG 2>&1 | tail -30 ;;
    rankshare:*) timeout 3000 python tools/rank_share.py $TAG "${step#rankshare:}" 2>&1 | tail -30 ;;
    cmd:*) bash -c "${step#cmd:}" ;;
    *) echo "unknown step $step"; rc=2 ;;
  esac
done
exit $rc
