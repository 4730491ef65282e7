#!/bin/bash
# kernel trace of staged steps under emp_set_option values: bash tools/r05_trace.sh <tag> [name=value ...]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; shift
mkdir -p gpurun_out/r05
rm -rf gpurun_out/r05/trace_$tag
timeout 300 rocprofv3 --kernel-trace -f csv -d gpurun_out/r05/trace_$tag -o t -- python tools/gap_probe.py "$@" > gpurun_out/r05/gap_probe_$tag.log 2>&1
tail -2 gpurun_out/r05/gap_probe_$tag.log
F=$(find gpurun_out/r05/trace_$tag -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $F 1500 26 | tee gpurun_out/r05/timeline_$tag.txt
head -1 $F > gpurun_out/r05/trace_${tag}_excerpt.csv; sed -n '1500,2100p' $F >> gpurun_out/r05/trace_${tag}_excerpt.csv
rm -rf gpurun_out/r05/trace_$tag
