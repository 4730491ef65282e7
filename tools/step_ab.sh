#!/bin/bash
# In-step A/B of edge-kernel options: bench.py --no-legs --no-cpu-baseline for each "name=value,name=value" argument.
# Usage: tools/step_ab.sh OUTDIR "edge_form=0,edge_block=128" "edge_form=1" ...
OUT=$1; shift
mkdir -p $OUT
for combo in "$@"; do
  args=""
  for kv in ${combo//,/ }; do args="$args --opt $kv"; done
  tag=${combo//,/_}; tag=${tag//=/}
  python bench.py --no-legs --no-cpu-baseline ${BENCH_ARGS:-} $args > $OUT/bench_$tag.json 2> $OUT/err_$tag.txt
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$tag.json"))
    k=d["kernels_ms"]
    print("%-44s ms/step %.4f  sweep frac %.3f (alone %.3f)  edge alone %.1f us  qp %.1f us" % ("$combo", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"], k["dp_edge"]*1e3, k["path_qp"]*1e3))
except Exception as e:
    print("$combo", "FAILED", e)
PY
done
