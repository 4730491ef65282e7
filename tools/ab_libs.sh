#!/bin/bash
# usage: ab_libs.sh OUT lib1 lib2 ... ; runs bench for each library variant (twice, alternating)
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
for lib in "$@"; do
  cp variants/$lib emplanner_carla_amd/libemplanner.so
  python bench.py --no-legs --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/b_${lib}_$rep.json 2> $OUT/e_${lib}_$rep.txt
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_${lib}_$rep.json")); k=d["kernels_ms"]
    print("%-28s ms/step %.4f  sweep frac %.3f  edge alone %.1f us  qp %.1f us" % ("$lib", d["ms_per_step"], d["roofline"]["frac"], k["dp_edge"]*1e3, k["path_qp"]*1e3))
except Exception as e:
    print("$lib FAILED", e)
PY
done
done
cp variants/lib_cur.so emplanner_carla_amd/libemplanner.so
