for b in 11 12; do echo "max bits $b"; PJB_RADIX_BITS=$b python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k={x['name']:x for x in d['kernels']}
print(round(d['value']/1e9,3), 'Greads/s', round(d['ms_per_step'],3), 'ms/step kernels', d['device_kernel_ms_per_step'], 'passes', d['sort_passes'], 'scatter', k['rs_scatter']['avg_ms'], 'hist', k['rs_hist']['avg_ms'], 'scan', k['rs_scan_apply']['avg_ms'], k['rs_scan_reduce']['avg_ms'], 'k4', k['k4_pairs']['avg_ms'])
"; done
