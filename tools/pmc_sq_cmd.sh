#!/bin/bash
# SQ counter passes (as tools/pmc_sq.sh) for the kernels of an arbitrary python command.
# Usage: tools/pmc_sq_cmd.sh TAG script.py [args...]   -> gpurun_out/sq_TAG/summary.csv    (PASSES="1 2 3" selects passes)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
P[1]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P[2]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P[3]="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES"
for i in ${PASSES:-1 2}; do
  timeout 600 rocprofv3 --kernel-trace --pmc ${P[$i]} -f csv -d $OUT/p$i -o sq -- python "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (rc $?)" >> $OUT/failed.txt
done
python tools/summarize_sq.py $OUT
