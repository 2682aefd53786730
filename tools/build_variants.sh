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
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $flags -shared \
      -o tools/variants/libpjb_$name.so portcullis_amd/csrc/pjb_api.hip &
done
wait
ls -la tools/variants/
