#!/bin/bash
# gpu_dqn_recon.sh TAG -- on the MI355X box: the DQN tests and bench line with dZ2's image (FLY_DQN_DW2_RECON=0) and without it (=1)
TAG=${1:-rc}
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
FLY_DQN_DW2_RECON=1 step pytest1 500 python -m pytest tests/test_dqn.py tests/test_dqn_h2_gpu.py -m gpu -q -x > $OUT/tests_recon.log 2>&1; tail -5 $OUT/tests_recon.log
grep -q " passed" $OUT/tests_recon.log && ! grep -q " failed" $OUT/tests_recon.log || exit 1
FLY_DQN_DW2_RECON=0 step pytest0 500 python -m pytest tests/test_dqn.py tests/test_dqn_h2_gpu.py -m gpu -q -x > $OUT/tests_image.log 2>&1; tail -3 $OUT/tests_image.log
grep -q " passed" $OUT/tests_image.log && ! grep -q " failed" $OUT/tests_image.log || exit 1
for i in 1 2; do
  FLY_DQN_DW2_RECON=1 step bench1 300 python bench.py --workload dqn --steps 10 --warmup 2 > $OUT/bench_recon_$i.json 2> /dev/null
  FLY_DQN_DW2_RECON=0 step bench0 300 python bench.py --workload dqn --steps 10 --warmup 2 > $OUT/bench_image_$i.json 2> /dev/null
done
python3 - <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/*/bench_*_?.json")):
    pass
PY
for f in $OUT/bench_*_?.json; do python3 -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][0]); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], [k['avg_launch_us'] for k in d['kernels'][:2]])"; done
echo "recon done"
