#!/bin/bash
# gpu_ab_libs.sh TAG LIB... -- on the MI355X box: tools/time_fused.py (fused gradient, gradient + optimizer) with each library in turn, twice
# (LIB = a path under fly_bproject_amd/, e.g. libflyhip.so libflyhip_ab.so: A/B builds of one source tree, selected through FLYHIP_LIB)
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for i in 1 2; do
  for lib in "$@"; do
    FLYHIP_LIB=$PWD/fly_bproject_amd/$lib timeout -k 10 200 python tools/time_fused.py 40960 200 > $OUT/time_${lib%.so}_$i.txt 2>&1 || exit 1
    echo "$lib $i: $(grep -m1 f16x2 $OUT/time_${lib%.so}_$i.txt | cut -c1-90)"
  done
done
echo "ab libs done"
