#!/bin/bash
# gpu_check.sh TAG -- on the MI355X box: the whole GPU test suite, then the driver's bench line.
TAG=${1:-chk}
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
step pytest 800 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
step bench 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.json
echo "check done"
