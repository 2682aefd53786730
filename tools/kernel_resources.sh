#!/bin/bash
# Register / LDS / occupancy table of every kernel of libportcullis_amd.so (no GPU needed: hipcc's resource-usage remarks).
#   bash tools/kernel_resources.sh [name-filter]
cd "$(dirname "$0")/../portcullis_amd/csrc"
#   UNITS="pjb_api" bash tools/kernel_resources.sh k1_   (one translation unit only: pjb_api | pjb_extra_api | pjb_ingest_api)
: > /tmp/pjb_resource.txt
for u in ${UNITS:-pjb_api pjb_extra_api pjb_ingest_api}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Rpass-analysis=kernel-resource-usage -c -o /tmp/${u}_res.o $u.hip 2>> /tmp/pjb_resource.txt &
done
wait
python3 - "$1" <<'PY'
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
txt = open('/tmp/pjb_resource.txt').read()
print(f"{'kernel':58s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'occ':>4s} {'sspill':>6s} {'vspill':>6s} {'LDS':>6s}")
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].split(' [')[0].strip()
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    import subprocess
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0].replace('pjb::', '')
    if flt and flt not in dn:
        continue
    vals = [g('VGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g('SGPRs Spill'), g('VGPRs Spill'), g(r'LDS Size \[bytes/block\]')]
    print("%-58s %5d %5d %7d %4d %6d %6d %6d" % ((dn[:58],) + tuple(vals)))
PY
