#!/bin/bash
# Collect a few SQ counters for one command (development aid).  Usage: tools/pmc_probe.sh TAG "COUNTERS" -- cmd...
TAG=$1; CTR=$2; shift 3
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $CTR -f csv -d $OUT -o $TAG -- "$@" > $OUT/run.log 2>&1
python - "$OUT/${TAG}_counter_collection.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: f"{sum(x)/len(x):.4g}" for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
