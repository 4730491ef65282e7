#!/bin/bash
# Round-6 evidence run on the GPU box: bench lines and the rocprofv3 passes behind profiles/r06*.
# Afterwards, here: tools/r06_file_evidence.sh copies the results under profiles/ and refreshes profiles/counters.json.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06e
rm -rf gpurun_out/prof_r06* gpurun_out/sq_r06a gpurun_out/sq_r06c_cfg5
STEPS=20 WARMUP=5 timeout 400 bash tools/profile.sh r06a > gpurun_out/r06e/profile_r06a.log 2>&1; echo "r06a $?"
STEPS=30 WARMUP=5 BENCH_ARGS="--scenes-per-gpu 32768" timeout 400 bash tools/profile.sh r06b_32768 > gpurun_out/r06e/profile_r06b.log 2>&1; echo "r06b $?"
STEPS=10 WARMUP=3 BENCH_ARGS="--config cfg5" timeout 400 bash tools/profile.sh r06c_cfg5 > gpurun_out/r06e/profile_r06c.log 2>&1; echo "r06c $?"
timeout 300 bash tools/pmc_sq.sh r06a > gpurun_out/r06e/sq_r06a.log 2>&1; echo "sq a $?"
STEPS=8 WARMUP=2 timeout 400 bash tools/pmc_sq.sh r06c_cfg5 --config cfg5 > gpurun_out/r06e/sq_r06c.log 2>&1; echo "sq c $?"
# the counter passes first, summarised HERE into this copy's profiles/counters.json: the bench lines below read it (roofline.traffic,
# roofline_step, the committed trace's launch durations), so they quote the counters of the build they measure
python tools/summarize_profile.py r06b_32768 32768 30 5 > /dev/null
python tools/summarize_profile.py r06c_cfg5 4096 10 3 cfg5_120x21_16obs > /dev/null
python tools/summarize_profile.py r06a 4096 20 5 > /dev/null
python tools/summarize_sq.py gpurun_out/sq_r06c_cfg5 r06c_cfg5 cfg5_120x21_16obs 4096 corridor > /dev/null
python tools/summarize_sq.py gpurun_out/sq_r06a r06a cfg2_40x9_8obs 4096 corridor > /dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06e/bench_default_s20.json 2> gpurun_out/r06e/bench_default_s20.err
python bench.py > gpurun_out/r06e/bench_default.json 2> gpurun_out/r06e/bench_default.err
N="--no-cpu-baseline --no-legs"
python bench.py --pipeline staged $N > gpurun_out/r06e/bench_staged.json 2>/dev/null
python bench.py --pipeline staged --opt sweep_exclusive=2 $N > gpurun_out/r06e/bench_exclusive_sweep.json 2>/dev/null
python bench.py --opt edge_form=1 $N > gpurun_out/r06e/bench_lockstep_edge.json 2>/dev/null
python bench.py --opt path_qp_form=1 $N > gpurun_out/r06e/bench_path_qp_form1.json 2>/dev/null
python bench.py --start-ahead 2.0 $N > gpurun_out/r06e/bench_on_node_batch.json 2>/dev/null
python bench.py --no-pipeline $N > gpurun_out/r06e/bench_one_batch.json 2>/dev/null
python bench.py --pipeline 6 $N > gpurun_out/r06e/bench_lanes6.json 2>/dev/null
python bench.py --opt lane_edge_order=1 $N > gpurun_out/r06e/bench_lane_edge_order1.json 2>/dev/null
python bench.py --scenes-per-gpu 8192 --steps 50 $N > gpurun_out/r06e/bench_8192.json 2>/dev/null
python bench.py --scenes-per-gpu 8192 --steps 50 --opt lane_edge_order=0 $N > gpurun_out/r06e/bench_8192_unordered.json 2>/dev/null
python bench.py --scenes-per-gpu 32768 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r06e/bench_32768.json 2>/dev/null
python bench.py --scenes-per-gpu 32768 --steps 30 --warmup 5 --opt lane_edge_order=0 $N > gpurun_out/r06e/bench_32768_unordered.json 2>/dev/null
python bench.py --total-scenes 32768 --steps 30 --warmup 5 $N > gpurun_out/r06e/bench_total_32768_one_gpu.json 2>/dev/null
python bench.py --force-gather-path $N > gpurun_out/r06e/bench_gather_path.json 2>/dev/null
python bench.py --latency > gpurun_out/r06e/bench_latency.json 2>/dev/null
python bench.py --config cfg5 --steps 20 --warmup 3 > gpurun_out/r06e/bench_cfg5.json 2> gpurun_out/r06e/bench_cfg5.err
python bench.py --scene-dist survey --arcs survey $N > gpurun_out/r06e/bench_survey_tight.json 2>/dev/null
python bench.py --arcs survey $N > gpurun_out/r06e/bench_corridor_tight.json 2>/dev/null
python bench.py --scene-dist worst $N > gpurun_out/r06e/bench_worst.json 2>/dev/null
for q in 4 12; do for m in staged 3; do GPU_MAX_HW_QUEUES=$q python bench.py --pipeline $m $N > gpurun_out/r06e/bench_queues${q}_${m}.json 2>/dev/null; done; done
EMP_BENCH_BACKEND=gloo python bench.py --gpus 2 --scenes-per-gpu 2048 --steps 20 --no-cpu-baseline > gpurun_out/r06e/bench_bare_gpus2_gloo_one_gpu.json 2> gpurun_out/r06e/bench_bare_gpus2.err
python bench_dropin.py --requests 400 > gpurun_out/r06e/bench_dropin.json 2>/dev/null
python tools/host_rate_probe.py 4096 200 > gpurun_out/r06e/host_rate_probe.txt 2>&1
python tools/call_cost_probe.py > gpurun_out/r06e/call_cost_probe.txt 2>&1
for f in gpurun_out/r06e/bench_*.json; do echo "$f: $(cut -c1-160 $f)"; done
