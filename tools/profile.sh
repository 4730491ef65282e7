#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then two PMC passes (FETCH_SIZE, WRITE_SIZE) as the
# MI355X guide prescribes (separate passes; never combined with sys/runtime tracing; counter passes serialise the kernels and
# are slow, so they skip bench.py's untimed settling steps - counters do not depend on clocks).  Usage: [BENCH_ARGS=...] tools/profile.sh TAG
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
STEPS=${STEPS:-100}; WARMUP=${WARMUP:-10}   # bench.py's defaults
ARGS="bench.py --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-legs ${BENCH_ARGS:-}"   # BENCH_ARGS: e.g. "--scenes-per-gpu 32768", "--config cfg5"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o $TAG -- python $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o $TAG -- python $ARGS --settle-steps 0 > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o $TAG -- python $ARGS --settle-steps 0 > $OUT/bench_pmc_write.log 2>&1
python bench.py --steps $STEPS --warmup $WARMUP ${BENCH_ARGS:-} ${PLAIN_ARGS:-} > $OUT/bench_plain.json 2> $OUT/bench_plain.err
find $OUT -name "*.csv" | head -20
grep -h '"metric"' $OUT/bench_trace.log | cut -c1-200
