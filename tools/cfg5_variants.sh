#!/bin/bash
# Development A/B of BASELINE configs[4]'s bench line over library builds (variants/lib_*.so) and pipeline modes.
# Usage: tools/cfg5_variants.sh "mode mode ..."
cd "${GRAFT_REPO_ROOT:-/root/repo}"
MODES=${1:-auto}
cp emplanner_carla_amd/libemplanner.so /tmp/stock.so
run() {
  for m in $MODES; do python bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu-baseline --pipeline $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms']; print('  $m', d['ms_per_step'], d['value'], 'sweep', d['roofline']['frac'], 'edge', k['dp_edge'], 'st', k['speed_dp'], 'qp', k['path_qp'])"; done
}
echo "== stock"; run
for v in variants/lib_*.so; do echo "== $v"; cp $v emplanner_carla_amd/libemplanner.so; run; done
cp /tmp/stock.so emplanner_carla_amd/libemplanner.so
