#!/bin/bash
# Second half of tools/r02_evidence.sh (can be run alone): the survey-layout profile and the four SQ counter passes, each under its
# own time limit (counter passes serialise the kernels; one that hangs must not take the whole budget).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02
rm -rf gpurun_out/prof_r02d_survey gpurun_out/sq_r02a gpurun_out/sq_r02d_survey gpurun_out/sq_r02e_worst gpurun_out/sq_r02c_cfg5
date +%s > gpurun_out/r02/rest_start
STEPS=100 WARMUP=10 BENCH_ARGS="--scene-dist survey" timeout 150 bash tools/profile.sh r02d_survey > gpurun_out/r02/profile_r02d.log 2>&1; echo "r02d $?"
timeout 100 bash tools/pmc_sq.sh r02a > gpurun_out/r02/sq_r02a.log 2>&1; echo "sq a $?"
timeout 100 bash tools/pmc_sq.sh r02d_survey --scene-dist survey > gpurun_out/r02/sq_r02d.log 2>&1; echo "sq d $?"
timeout 100 bash tools/pmc_sq.sh r02e_worst --scene-dist worst > gpurun_out/r02/sq_r02e.log 2>&1; echo "sq e $?"
STEPS=8 WARMUP=2 timeout 120 bash tools/pmc_sq.sh r02c_cfg5 --config cfg5 > gpurun_out/r02/sq_r02c.log 2>&1; echo "sq c $?"
echo elapsed $(( $(date +%s) - $(cat gpurun_out/r02/rest_start) ))
