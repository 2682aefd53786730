cd $GRAFT_REPO_ROOT
for k in 1 2 3; do for who in r05 now; do
  exe=portcullis_amd/host/portcullis_amd; [ $who = r05 ] && exe=tools/variants/r05/host/portcullis_amd
  ( time PJB_PROFILE_HOST=1 $exe junc -t $(nproc) --orientation FR -o /tmp/pjb_bench_e2e/prof/p_$who /tmp/pjb_bench_e2e/prep ) > gpurun_out/e2eprof_${who}_$k.txt 2>&1
  echo "== $who run $k"; grep -E "^real|t=[0-9.]+ s: (device thread: context|workers and device|outputs)|device thread: alive|main: " gpurun_out/e2eprof_${who}_$k.txt | cut -c1-230
done; done
