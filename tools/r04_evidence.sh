#!/bin/bash
# Round-4 evidence run on the GPU box: bench lines and the rocprofv3 passes behind profiles/r04*.
# Afterwards, here: tools/r04_file_evidence.sh copies the results under profiles/ and refreshes profiles/counters.json.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04e
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04e/bench_default_s20.json 2> gpurun_out/r04e/bench_default_s20.err
python bench.py > gpurun_out/r04e/bench_default.json 2> gpurun_out/r04e/bench_default.err
python bench.py --opt sweep_exclusive=0 --no-cpu-baseline --no-legs > gpurun_out/r04e/bench_overlapped_sweep.json 2>/dev/null
python bench.py --opt sweep_exclusive=1 --no-cpu-baseline --no-legs > gpurun_out/r04e/bench_exclusive_sweep.json 2>/dev/null
python bench.py --no-pipeline --no-cpu-baseline --no-legs > gpurun_out/r04e/bench_one_batch.json 2>/dev/null
python bench.py --pipeline 3 --no-cpu-baseline > gpurun_out/r04e/bench_lanes3.json 2>/dev/null
python bench.py --force-gather-path --no-cpu-baseline > gpurun_out/r04e/bench_gather_path.json 2>/dev/null
python bench.py --latency > gpurun_out/r04e/bench_latency.json 2>/dev/null
python bench.py --config cfg5 --steps 20 --warmup 3 > gpurun_out/r04e/bench_cfg5.json 2> gpurun_out/r04e/bench_cfg5.err
python bench.py --scenes-per-gpu 32768 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04e/bench_32768.json 2>/dev/null
rm -rf gpurun_out/prof_r04* gpurun_out/sq_r04*
STEPS=20 WARMUP=5 timeout 300 bash tools/profile.sh r04a > gpurun_out/r04e/profile_r04a.log 2>&1; echo "r04a $?"
STEPS=30 WARMUP=5 BENCH_ARGS="--scenes-per-gpu 32768" timeout 300 bash tools/profile.sh r04b_32768 > gpurun_out/r04e/profile_r04b.log 2>&1; echo "r04b $?"
STEPS=10 WARMUP=3 BENCH_ARGS="--config cfg5" timeout 300 bash tools/profile.sh r04c_cfg5 > gpurun_out/r04e/profile_r04c.log 2>&1; echo "r04c $?"
timeout 200 bash tools/pmc_sq.sh r04a > gpurun_out/r04e/sq_r04a.log 2>&1; echo "sq a $?"
STEPS=8 WARMUP=2 timeout 240 bash tools/pmc_sq.sh r04c_cfg5 --config cfg5 > gpurun_out/r04e/sq_r04c.log 2>&1; echo "sq c $?"
for f in gpurun_out/r04e/bench_*.json; do echo "$f: $(cut -c1-160 $f)"; done
