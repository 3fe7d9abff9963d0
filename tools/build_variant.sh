#!/bin/bash
# build_variant.sh NAME "-DFLAG ..." : A/B build of the kernels into build_ab/libflyhip_NAME.so
set -e
cd "$(dirname "$0")/../fly_bproject_amd/csrc"
mkdir -p ../../build_ab/$1
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I../../include -Wno-unused-function $2 -c $f -o ../../build_ab/$1/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/libflyhip_$1.so ../../build_ab/$1/*.o
echo built build_ab/libflyhip_$1.so
