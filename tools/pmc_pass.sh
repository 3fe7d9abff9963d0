#!/bin/bash
# pmc_pass.sh TAG "COUNTER ..." [env assignments for the program]  -- one rocprofv3 PMC pass of tools/prof_kernels.py
# (counters in their own run: never together with trace domains), summarised per kernel into gpurun_out/TAG.csv
set -e
TAG=$1; shift
CTRS=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 tools/prof_kernels.py > $OUT/run.log 2>&1
python3 - "$OUT" "gpurun_out/$TAG.csv" <<'PY'
import csv, glob, re, sys, collections
out, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+)(<[^(]*>)?\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        k = ((m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
with open(dst, "w") as fo:
    fo.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for (k, c), (v, n) in sorted(acc.items()):
        fo.write("%s,%s,%d,%.6g\n" % (k, c, n, v / n))
print(open(dst).read())
PY
find $OUT -name "*.csv" -size +2M -delete
