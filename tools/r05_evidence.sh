#!/bin/bash
# Round-5 evidence run on the GPU box: bench lines and the rocprofv3 passes behind profiles/r05*.
# Afterwards, here: tools/r05_file_evidence.sh copies the results under profiles/ and refreshes profiles/counters.json.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05e
rm -rf gpurun_out/prof_r05* gpurun_out/sq_r05a gpurun_out/sq_r05c_cfg5
STEPS=20 WARMUP=5 timeout 300 bash tools/profile.sh r05a > gpurun_out/r05e/profile_r05a.log 2>&1; echo "r05a $?"
STEPS=30 WARMUP=5 BENCH_ARGS="--scenes-per-gpu 32768" timeout 300 bash tools/profile.sh r05b_32768 > gpurun_out/r05e/profile_r05b.log 2>&1; echo "r05b $?"
STEPS=10 WARMUP=3 BENCH_ARGS="--config cfg5" timeout 300 bash tools/profile.sh r05c_cfg5 > gpurun_out/r05e/profile_r05c.log 2>&1; echo "r05c $?"
timeout 240 bash tools/pmc_sq.sh r05a > gpurun_out/r05e/sq_r05a.log 2>&1; echo "sq a $?"
STEPS=8 WARMUP=2 timeout 300 bash tools/pmc_sq.sh r05c_cfg5 --config cfg5 > gpurun_out/r05e/sq_r05c.log 2>&1; echo "sq c $?"
# the counter passes first, summarised HERE into this copy's profiles/counters.json: the bench lines below read it (roofline.traffic,
# roofline_step), so they quote the counters of the build they measure; tools/r05_file_evidence.sh repeats the summaries at home
python tools/summarize_profile.py r05b_32768 32768 30 5 > /dev/null
python tools/summarize_profile.py r05c_cfg5 4096 10 3 cfg5_120x21_16obs > /dev/null
python tools/summarize_profile.py r05a 4096 20 5 > /dev/null
python tools/summarize_sq.py gpurun_out/sq_r05c_cfg5 r05c_cfg5 cfg5_120x21_16obs 4096 corridor > /dev/null
python tools/summarize_sq.py gpurun_out/sq_r05a r05a cfg2_40x9_8obs 4096 corridor > /dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r05e/bench_default_s20.json 2> gpurun_out/r05e/bench_default_s20.err
python bench.py > gpurun_out/r05e/bench_default.json 2> gpurun_out/r05e/bench_default.err
python bench.py --pipeline staged --opt sweep_exclusive=2 --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_exclusive_sweep.json 2>/dev/null
python bench.py --opt edge_form=1 --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_lockstep_edge.json 2>/dev/null
python bench.py --start-ahead 2.0 --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_on_node_batch.json 2>/dev/null
python bench.py --no-pipeline --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_one_batch.json 2>/dev/null
python bench.py --pipeline staged --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_staged.json 2>/dev/null
python bench.py --pipeline 6 --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_lanes6.json 2>/dev/null
python bench.py --opt lane_edge_order=1 --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_lane_edge_order.json 2>/dev/null
python bench.py --opt lane_edge_order=1 --scenes-per-gpu 32768 --steps 30 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_32768_lane_edge_order.json 2>/dev/null
python bench.py --force-gather-path --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_gather_path.json 2>/dev/null
python bench.py --latency > gpurun_out/r05e/bench_latency.json 2>/dev/null
python bench.py --config cfg5 --steps 20 --warmup 3 > gpurun_out/r05e/bench_cfg5.json 2> gpurun_out/r05e/bench_cfg5.err
python bench.py --scenes-per-gpu 32768 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r05e/bench_32768.json 2>/dev/null
python bench.py --scene-dist survey --arcs survey --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_survey_tight.json 2>/dev/null
python bench.py --arcs survey --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_corridor_tight.json 2>/dev/null
python bench.py --scene-dist worst --no-cpu-baseline --no-legs > gpurun_out/r05e/bench_worst.json 2>/dev/null
python tools/host_rate_probe.py 4096 200 > gpurun_out/r05e/host_rate_probe.txt 2>&1
python tools/pcie_probe.py > gpurun_out/r05e/pcie_probe.txt 2>&1
for f in gpurun_out/r05e/bench_*.json; do echo "$f: $(cut -c1-160 $f)"; done
