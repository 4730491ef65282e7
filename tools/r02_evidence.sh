#!/bin/bash
# Round-2 evidence run on the GPU box: every bench line and the rocprofv3 passes behind profiles/r02*.
# Afterwards, here: tools/r02_file_evidence.sh copies the results under profiles/ and refreshes profiles/counters.json.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err
python bench.py --alt-pipeline 3 --no-cpu-baseline > gpurun_out/r02/bench_default_and_lanes3.json 2>/dev/null
python bench.py --no-pipeline --no-cpu-baseline > gpurun_out/r02/bench_one_batch.json 2>/dev/null
for n in 2 3 4; do python bench.py --pipeline $n --no-cpu-baseline > gpurun_out/r02/bench_lanes$n.json 2>/dev/null; done
python bench.py --force-gather-path --no-cpu-baseline > gpurun_out/r02/bench_gather_path.json 2>/dev/null
python bench.py --force-gather-path --records trajectory --no-cpu-baseline > gpurun_out/r02/bench_gather_path_trajectory.json 2>/dev/null
python bench.py --dp-mode fused --no-cpu-baseline > gpurun_out/r02/bench_fused.json 2>/dev/null
python bench.py --scene-dist survey --cpu-pool 0 --cpu-sample 24 > gpurun_out/r02/bench_survey.json 2>/dev/null
python bench.py --scene-dist worst --cpu-pool 0 --cpu-sample 24 > gpurun_out/r02/bench_worst.json 2>/dev/null
python bench.py --latency > gpurun_out/r02/bench_latency.json 2>/dev/null
python bench.py --config cfg5 --steps 20 --warmup 3 > gpurun_out/r02/bench_cfg5.json 2> gpurun_out/r02/bench_cfg5.err
python bench.py --config cfg5 --steps 20 --warmup 3 --pipeline off --no-cpu-baseline > gpurun_out/r02/bench_cfg5_one_batch.json 2>/dev/null
python bench.py --scenes-per-gpu 32768 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_32768.json 2>/dev/null
rm -rf gpurun_out/prof_r02* gpurun_out/sq_r02*
STEPS=100 WARMUP=10 bash tools/profile.sh r02a > gpurun_out/r02/profile_r02a.log 2>&1
STEPS=30 WARMUP=5 BENCH_ARGS="--scenes-per-gpu 32768" bash tools/profile.sh r02b_32768 > gpurun_out/r02/profile_r02b.log 2>&1
STEPS=10 WARMUP=3 BENCH_ARGS="--config cfg5" bash tools/profile.sh r02c_cfg5 > gpurun_out/r02/profile_r02c.log 2>&1
bash tools/r02_evidence_counters.sh      # the survey profile and the SQ passes, each under its own time limit
for f in gpurun_out/r02/bench_*.json; do echo "$f: $(cut -c1-230 $f)"; done
