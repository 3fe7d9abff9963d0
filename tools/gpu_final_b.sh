#!/bin/bash
# gpu_final_b.sh TAG -- second half of tools/gpu_final.sh: the rocprofv3 passes (kernel-trace stats, PMC traffic / SQ counters, bench under the profiler)
TAG=${1:-r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 bash tools/collect_profiles.sh $TAG/prof > $OUT/collect.log 2>&1
echo "[profiles] rc=$?" >&2
tail -3 $OUT/collect.log
echo "final b done"
