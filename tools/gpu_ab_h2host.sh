#!/bin/bash
# gpu_ab_h2host.sh TAG -- on the MI355X box: the fp16x2 optimizer step with layer 1's epilogues hosted in MFMA gaps (libflyhip.so) against the
# build without (fly_bproject_amd/libflyhip_ab.so, -DFS_H2_HOST_FWD=0): tests of the new build, then timings of both, alternating.
TAG=${1:-ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
step() {
    local name=$1 secs=$2; shift 2
    timeout -k 10 "$secs" "$@"
    local rc=$?
    echo "[$name] rc=$rc" >&2
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out: stopping" >&2; exit $rc; fi
    return $rc
}
step pytest 600 python -m pytest tests/test_fused_h2_gpu.py tests/test_mlp_train_gpu.py tests/test_dqn_h2_gpu.py -m gpu -q -x > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
grep -q " passed" $OUT/tests.log && ! grep -q " failed" $OUT/tests.log || exit 1
for i in 1 2; do
  step time_new 200 python tools/time_fused.py 40960 200 > $OUT/time_host_$i.txt 2>&1
  FLYHIP_LIB=$PWD/fly_bproject_amd/libflyhip_ab.so step time_old 200 python tools/time_fused.py 40960 200 > $OUT/time_nohost_$i.txt 2>&1
done
step stamp 200 python tools/stamp_fused.py 40960 f16x2 > $OUT/stamps_host.txt 2>&1
FLYHIP_LIB=$PWD/fly_bproject_amd/libflyhip_ab.so step stamp_old 200 python tools/stamp_fused.py 40960 f16x2 > $OUT/stamps_nohost.txt 2>&1
step bench 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_dqn --no_alt_gemm > $OUT/bench_host.json 2> /dev/null
FLYHIP_LIB=$PWD/fly_bproject_amd/libflyhip_ab.so step bench_old 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_dqn --no_alt_gemm > $OUT/bench_nohost.json 2> /dev/null
grep -h "f16x2" $OUT/time_*.txt
echo "ab done"
