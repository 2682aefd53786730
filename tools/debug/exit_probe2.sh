/opt/rocm/bin/hipcc -O2 -Wno-unused-value -o /tmp/exit_probe tools/debug/exit_probe.cc -lpthread || exit 1
for cfg in "0 2 exit" "0 2 free" "0 2 pfree" "0 2 exit" "0 2 free" "0 2 pfree" "0 1 exit" "0 1 free" "0 1 pfree"; do
  sleep 2
  s=$(date +%s.%N); out=$(/tmp/exit_probe $cfg); e=$(date +%s.%N)
  python3 -c "print('exit_probe $cfg: $out wall %.3f s' % ($e - $s))"
done | tee gpurun_out/r06_exit_probe2.txt
