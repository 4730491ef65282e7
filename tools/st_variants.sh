#!/bin/bash
# Development A/B of the S-T speed DP kernel: stock build, its previous-round kernel (EMP_ST_DP_V1), and every
# variants/lib_*.so through EMP_DBG_LIB.  Usage: tools/st_variants.sh [B ...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SIZES=${@:-4096}
for B in $SIZES; do
  echo "== stock B=$B"; python tools/st_microbench.py $B 2>&1 | grep -v "^terminal"
  if [ -n "$EMP_ST_HAS_V1" ]; then echo "== v1 kernel B=$B"; EMP_ST_DP_V1=1 python tools/st_microbench.py $B 2>&1 | grep -v "^terminal"; fi
  for v in variants/lib_*.so; do echo "== $v B=$B"; EMP_DBG_LIB=$v python tools/st_microbench.py $B 2>&1 | grep -v "^terminal"; done
done
