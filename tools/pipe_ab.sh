#!/bin/bash
# bench.py --no-legs --no-cpu-baseline under each --pipeline value: tools/pipe_ab.sh OUTDIR staged 2 3 4 ...
OUT=$1; shift
mkdir -p $OUT
for p in "$@"; do
  python bench.py --no-legs --no-cpu-baseline --pipeline $p ${BENCH_ARGS:-} > $OUT/bench_p$p.json 2> $OUT/err_p$p.txt
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_p$p.json")); k=d["kernels_ms"]
    print("pipeline %-8s ms/step %.4f  value %.3e  sweep frac %.3f (alone %.3f)" % ("$p", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["frac_alone"]))
except Exception as e:
    print("$p FAILED", e)
PY
done
