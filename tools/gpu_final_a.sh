#!/bin/bash
# gpu_final_a.sh TAG -- first half of tools/gpu_final.sh (a call is limited to 20 minutes): tests, bench lines, stamps, timings, curves.
TAG=${1:-r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
step() {
    local name=$1 secs=$2; shift 2
    timeout -k 10 "$secs" "$@"
    local rc=$?
    echo "[$name] rc=$rc" >&2
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out: stopping" >&2; exit $rc; fi
    return $rc
}
step pytest 800 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
step bench 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
step bench4096 200 python bench.py --num_envs 4096 --steps 5 --warmup 2 --no_cpu_baseline --no_dqn --no_alt_gemm > $OUT/bench_4096_envs.json 2> /dev/null
step bench16384 200 python bench.py --num_envs 16384 --steps 5 --warmup 2 --no_cpu_baseline --no_dqn --no_alt_gemm > $OUT/bench_16384_envs.json 2> /dev/null
step benchdqn 300 python bench.py --workload dqn --steps 10 --warmup 2 > $OUT/bench_dqn_f16x2.json 2> /dev/null
step benchdqn_b3 300 python bench.py --workload dqn --steps 10 --warmup 2 --dqn_gemm bf16x3 > $OUT/bench_dqn_bf16x3.json 2> /dev/null
step stampdqn 200 python tools/stamp_dqn.py 16 f16x2 > $OUT/stamps_dqn_f16x2.txt 2>&1
step stamp 200 python tools/stamp_fused.py 40960 f16x2 > $OUT/stamps_h2.txt 2>&1
step timef 200 python tools/time_fused.py 40960 200 > $OUT/time_fused.txt 2>&1
step curve_h2 300 python tools/train_curve.py 200 8192 hip bigGrav f16x2 > $OUT/training_curve_f16x2.txt 2>&1
step curve_b3 300 python tools/train_curve.py 200 8192 hip bigGrav bf16x3 > $OUT/training_curve_bf16x3.txt 2>&1
echo "final a done"
