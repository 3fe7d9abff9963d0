#!/bin/bash
# gpu_ab_dqn_libs.sh TAG LIB... -- on the MI355X box: the DQN tests with the first library, then the DQN bench line with each library in turn, twice
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
FLYHIP_LIB=$PWD/fly_bproject_amd/$1 timeout -k 10 500 python -m pytest tests/test_dqn.py tests/test_dqn_h2_gpu.py -m gpu -q -x > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
grep -q " passed" $OUT/tests.log && ! grep -q " failed" $OUT/tests.log || exit 1
for i in 1 2; do
  for lib in "$@"; do
    FLYHIP_LIB=$PWD/fly_bproject_amd/$lib timeout -k 10 300 python bench.py --workload dqn --steps 10 --warmup 2 > $OUT/bench_${lib%.so}_$i.json 2> /dev/null || exit 1
    python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench_${lib%.so}_$i.json') if l.startswith('{')][0]); print('$lib $i', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], [k['avg_launch_us'] for k in d['kernels'][:2]])"
  done
done
echo "ab dqn libs done"
