#!/bin/bash
# Development A/B of the S-T speed DP kernel: stock build and every variants/lib_*.so (built here with extra -D flags)
# through EMP_DBG_LIB.  Usage: tools/st_variants.sh [B ...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SIZES=${@:-4096}
for B in $SIZES; do
  echo "== stock B=$B"; python tools/st_microbench.py $B 2>&1 | grep -v "^terminal"
  for v in variants/lib_*.so; do echo "== $v B=$B"; EMP_DBG_LIB=$v python tools/st_microbench.py $B 2>&1 | grep -v "^terminal"; done
done
