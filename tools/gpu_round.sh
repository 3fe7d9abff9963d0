#!/bin/bash
# gpu_round.sh TAG [pytest args]: on the MI355X box -- the GPU tests, a default bench line, then the profile collection,
# each under its own timeout; a step that TIMES OUT (rc 124 / 137) ends the script (no GPU step after a hang).
TAG=${1:-r}
shift || true
mkdir -p gpurun_out
step() {   # step NAME SECONDS cmd...
    local name=$1 secs=$2; shift 2
    timeout -k 10 "$secs" "$@"
    local rc=$?
    echo "[$name] rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out: stopping"; exit $rc; fi
    return $rc
}
step pytest 700 python -m pytest tests -m gpu -q "$@" > gpurun_out/${TAG}_tests.log 2>&1
tail -4 gpurun_out/${TAG}_tests.log
step bench 300 python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.err
