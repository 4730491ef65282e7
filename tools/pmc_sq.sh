#!/bin/bash
# SQ counter passes for the kernels of one bench.py run (one batch in flight, so every kernel runs alone).
# Each pass is its own rocprofv3 run with --kernel-trace only (never combined with sys/runtime tracing).
# Usage: tools/pmc_sq.sh TAG [bench args...]   -> gpurun_out/sq_TAG/{p1,p2,p3,p4}/..., gpurun_out/sq_TAG/summary.csv
TAG=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
ARGS="bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-3} --no-cpu-baseline --no-legs --no-pipeline --settle-steps 0 $*"
[ -f $OUT/counters_available.txt ] || rocprofv3 -L > $OUT/counters_available.txt 2>&1
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P3="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES"
P4="GRBM_GUI_ACTIVE GRBM_COUNT"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P -f csv -d $OUT/p$i -o sq -- python $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed (rc $?)" >> $OUT/failed.txt
done
python tools/summarize_sq.py $OUT ${RECORD:-}
