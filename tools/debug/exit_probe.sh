# (experiment, round 6) the wall of a process that holds G GB of device memory, against what it spent inside main
/opt/rocm/bin/hipcc -O2 -Wno-unused-value -o /tmp/exit_probe tools/debug/exit_probe.cc -lpthread || exit 1
for cfg in ${CFGS:-"0 0 exit" "100 0 exit" "100 0 free" "100 0 big" "100 0 bigfree" "0 2 exit" "0 2 free" "100 2 exit" "0 0 exit" "100 0 exit" "100 0 free"}; do
  sleep 3
  s=$(date +%s.%N); out=$(/tmp/exit_probe $cfg); e=$(date +%s.%N)
  python3 -c "print('exit_probe $cfg: $out wall %.3f s' % ($e - $s))"
done | tee gpurun_out/r06_exit_probe${SUFFIX:-}.txt
