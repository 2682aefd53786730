/opt/rocm/bin/hipcc -O2 -Wno-unused-value -o /tmp/exit_probe tools/debug/exit_probe.cc -lpthread || exit 1
for cfg in "0:0" "8:0" "24:0" "24:1" "48:0" "0:0" "24:0"; do
  sleep 2
  s=$(date +%s.%N); out=$(env EXIT_PROBE_STREAMS=${cfg%%:*} $( [ ${cfg##*:} = 1 ] && echo EXIT_PROBE_DESTROY=1 ) /tmp/exit_probe 1 0 exit); e=$(date +%s.%N)
  python3 -c "print('exit_probe streams:destroy $cfg: $out wall %.3f s' % ($e - $s))"
done | tee gpurun_out/r06_exit_probe3.txt
