#!/bin/bash
# Round-3 evidence run on the GPU box (final build of round 3; kept as the record of how profiles/r03* were made - the EMP_* environment
# switches below were development switches of THAT build; since round 4 they are `bench.py --opt name=value`, tools/r04_evidence.sh).
# Afterwards, here: tools/r03_file_evidence.sh copies the results under profiles/ and refreshes profiles/counters.json.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
python bench.py --alt-pipeline 3 --no-cpu-baseline > gpurun_out/r03/bench_default_and_lanes3.json 2>/dev/null
python bench.py --no-pipeline --no-cpu-baseline > gpurun_out/r03/bench_one_batch.json 2>/dev/null
python bench.py --pipeline 3 --no-cpu-baseline > gpurun_out/r03/bench_lanes3.json 2>/dev/null
EMP_PATH_QP_PAIR=1 EMP_EDGE_BLOCK=256 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_round2_qp_and_edge_blocks.json 2>/dev/null
EMP_PATH_QP_PAIR=1 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_pair_qp.json 2>/dev/null
EMP_EDGE_BLOCK=256 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_edge_block_256.json 2>/dev/null
python bench.py --force-gather-path --no-cpu-baseline > gpurun_out/r03/bench_gather_path.json 2>/dev/null
python bench.py --force-gather-path --records trajectory --no-cpu-baseline > gpurun_out/r03/bench_gather_path_trajectory.json 2>/dev/null
python bench.py --dp-mode fused --no-cpu-baseline > gpurun_out/r03/bench_fused.json 2>/dev/null
python bench.py --scene-dist survey --cpu-pool 0 --cpu-sample 24 > gpurun_out/r03/bench_survey.json 2>/dev/null
python bench.py --scene-dist worst --cpu-pool 0 --cpu-sample 24 > gpurun_out/r03/bench_worst.json 2>/dev/null
python bench.py --latency > gpurun_out/r03/bench_latency.json 2>/dev/null
python bench.py --config cfg5 --steps 20 --warmup 3 > gpurun_out/r03/bench_cfg5.json 2> gpurun_out/r03/bench_cfg5.err
python bench.py --config cfg5 --steps 20 --warmup 3 --pipeline off --no-cpu-baseline > gpurun_out/r03/bench_cfg5_one_batch.json 2>/dev/null
python bench.py --scenes-per-gpu 32768 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_32768.json 2>/dev/null
rm -rf gpurun_out/prof_r03* gpurun_out/sq_r03*
STEPS=100 WARMUP=10 timeout 300 bash tools/profile.sh r03a > gpurun_out/r03/profile_r03a.log 2>&1; echo "r03a $?"
STEPS=30 WARMUP=5 BENCH_ARGS="--scenes-per-gpu 32768" timeout 300 bash tools/profile.sh r03b_32768 > gpurun_out/r03/profile_r03b.log 2>&1; echo "r03b $?"
STEPS=10 WARMUP=3 BENCH_ARGS="--config cfg5" timeout 300 bash tools/profile.sh r03c_cfg5 > gpurun_out/r03/profile_r03c.log 2>&1; echo "r03c $?"
timeout 200 bash tools/pmc_sq.sh r03a > gpurun_out/r03/sq_r03a.log 2>&1; echo "sq a $?"
STEPS=8 WARMUP=2 timeout 240 bash tools/pmc_sq.sh r03c_cfg5 --config cfg5 > gpurun_out/r03/sq_r03c.log 2>&1; echo "sq c $?"
for f in gpurun_out/r03/bench_*.json; do echo "$f: $(cut -c1-200 $f)"; done
