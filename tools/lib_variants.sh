#!/bin/bash
# Development A/B of whole-library builds: every variants/lib_*.so (built here with extra -D flags) takes the place of
# emplanner_carla_amd/libemplanner.so on the GPU box in turn; the stock build runs first and last.
# Usage: tools/lib_variants.sh "mode mode ..." [bench args]     (modes: off staged 2 3 ...)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
MODES=${1:-staged}; shift
cp emplanner_carla_amd/libemplanner.so /tmp/stock.so
run() {
  for m in $MODES; do for r in 1 2; do python bench.py --steps 200 --warmup 10 --no-cpu-baseline --pipeline $m "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  $m', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['mean_launch_us'], d['kernels_ms']['dp_edge'], d['kernels_ms']['path_qp'])"; done; done
}
echo "== stock"; run "$@"
for v in variants/lib_*.so; do
  echo "== $v"; cp $v emplanner_carla_amd/libemplanner.so; run "$@"
done
cp /tmp/stock.so emplanner_carla_amd/libemplanner.so
echo "== stock again"; run "$@"
