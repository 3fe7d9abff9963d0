#!/bin/bash
# collect_profiles.sh TAG — on the MI355X box: kernel-trace stats and separate PMC passes of
# tools/prof_kernels.py plus kernel stats of a short bench run, under gpurun_out/TAG/.
# Summarise afterwards with: python tools/summarize_pmc.py TAG gpurun_out/TAG/stats gpurun_out/TAG/pmc_*
set -e
TAG=${1:-prof}
shift || true
for kv in "$@"; do export "$kv"; done      # e.g. FLY_GEMM=bf16x3
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/prof_kernels.py > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_kernels.py > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_kernels.py > $OUT/pmc_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq1 -- python3 tools/prof_kernels.py > $OUT/pmc_sq1.log 2>&1
echo "sq1 done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq2 -- python3 tools/prof_kernels.py > $OUT/pmc_sq2.log 2>&1
echo "sq2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline > $OUT/bench_stats.log 2>&1
echo "bench stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dqn_stats -- python3 bench.py --workload dqn --steps 3 --warmup 1 --dqn_mini_batch 16 --kernel_reps 10 > $OUT/dqn_stats.log 2>&1
echo "dqn bench stats done"
# keep only the small per-kernel files (traces are large)
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
