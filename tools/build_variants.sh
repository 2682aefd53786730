#!/bin/bash
# Builds variants of libportcullis_amd.so for kernel A/B runs on the GPU box:
#   bash tools/build_variants.sh NAME1="-DFOO=1 -DBAR=2" NAME2="-DFOO=3" ...
# -> tools/variants/libpjb_NAME.so (git-ignored; travels with gpurun).  Use with PJB_LIB_PATH=tools/variants/libpjb_NAME.so
# (portcullis_amd/ffi.py; the C++ host layer is linked against the default build and is not affected).
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  echo "building $name: $flags"
  ( F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $flags"
    for u in pjb_api pjb_extra_api pjb_ingest_api; do /opt/rocm/bin/hipcc $F -c -o tools/variants/${name}_$u.o portcullis_amd/csrc/$u.hip & done; wait
    /opt/rocm/bin/hipcc $F -shared -o tools/variants/libpjb_$name.so tools/variants/${name}_pjb_api.o tools/variants/${name}_pjb_extra_api.o tools/variants/${name}_pjb_ingest_api.o
    rm -f tools/variants/${name}_*.o ) &
done
wait
ls -la tools/variants/
