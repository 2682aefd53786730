# (experiment, behind a `bench` step: the prepared directory of the bench's e2e leg) several device threads -- contexts -- on ONE GPU
P=/tmp/pjb_bench_e2e
TIMEFORMAT="%R s"
for n in 1 2 3 2 1; do
  for k in 1 2 3; do
    echo -n "devices $n: "
    time ( PORTCULLIS_DEVICES_SHARE_GPU=1 portcullis_amd/host/portcullis_amd junc -t $(nproc) --devices $n --orientation FR -o $P/dev$n/pc $P/prep > /dev/null 2>$P/err.txt || tail -2 $P/err.txt )
    md5sum $P/dev$n/pc.junctions.tab | cut -c1-12
  done
done
