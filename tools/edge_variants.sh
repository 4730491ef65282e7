#!/bin/bash
# Development A/B of the edge-cost kernel: EMP_EDGE_VARIANT (0 round-1 scan, 1 dense scan + box test), EMP_EDGE_PERSIST
# (1 persistent grid with work queue, 0 one block per item), EMP_EDGE_NC (columns per work item).
# Usage: tools/edge_variants.sh "variant:persist:nc ..."
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for cfg in ${1:-1:1:0 1:0:0}; do
  IFS=: read v pz nc <<< "$cfg"
  echo "== variant $v persist $pz nc $nc"
  export EMP_EDGE_VARIANT=$v EMP_EDGE_PERSIST=$pz EMP_EDGE_NC=$nc
  python tools/dp_microbench.py 4096 2>&1 | grep "mode 1"
  python tools/dp_microbench.py 4096 noobs 2>&1 | grep "mode 1"
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('one batch', d['ms_per_step'], d['kernels_ms'])"
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('pipelined', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done
