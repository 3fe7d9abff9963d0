#!/bin/bash
# gpu_dqn_h2.sh TAG -- on the MI355X box: the DQN update in both arithmetics: its tests, then the DQN bench line of each.
TAG=${1:-dq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
step() {   # step NAME SECONDS cmd...
    local name=$1 secs=$2; shift 2
    timeout -k 10 "$secs" "$@"
    local rc=$?
    echo "[$name] rc=$rc" >&2
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out: stopping" >&2; exit $rc; fi
    return $rc
}
step pytest 500 python -m pytest tests/test_dqn.py tests/test_dqn_h2_gpu.py -m gpu -q -x -s > $OUT/tests.log 2>&1; tail -15 $OUT/tests.log
grep -q "passed" $OUT/tests.log || exit 1
step benchdqn_h2 300 python bench.py --workload dqn --steps 10 --warmup 2 > $OUT/bench_dqn_f16x2.json 2> $OUT/bench_dqn_f16x2.err
step benchdqn_b3 300 python bench.py --workload dqn --steps 10 --warmup 2 --dqn_gemm bf16x3 > $OUT/bench_dqn_bf16x3.json 2> $OUT/bench_dqn_bf16x3.err
tail -c 1500 $OUT/bench_dqn_f16x2.json; tail -c 600 $OUT/bench_dqn_f16x2.err
echo "dqn h2 done"
