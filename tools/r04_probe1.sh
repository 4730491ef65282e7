#!/bin/bash
# Round 4, first GPU call: the whole GPU suite on the refactored library (options instead of environment switches), then
# the evidence for "what slows the sweep in the staged step": kernel trace of staged steps + the in-kernel clock probe
# under the candidate remedies.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04/pytest_gpu.log; cat gpurun_out/r04/pytest_gpu.log
timeout 300 python tools/step_variants.py --steps 100 --json gpurun_out/r04/step_variants1.json default off sweep_exclusive=1 back_stream_cus=128 back_stream_cus=192 back_stream_cus=64 sweep_exclusive=1,back_stream_cus=128 2>&1 | tail -20
rm -rf gpurun_out/r04/trace_staged
timeout 300 rocprofv3 --kernel-trace -f csv -d gpurun_out/r04/trace_staged -o t -- python tools/gap_probe.py > gpurun_out/r04/gap_probe.log 2>&1
F=$(find gpurun_out/r04/trace_staged -name "*kernel_trace.csv" | head -1)
python tools/sweep_window.py $F 60 gpurun_out/r04/sweep_window.json
python tools/timeline.py $F 1500 30
# keep the trace small: 80 steps of it
head -1 $F > gpurun_out/r04/trace_staged_excerpt.csv; sed -n '1500,2100p' $F >> gpurun_out/r04/trace_staged_excerpt.csv
rm -rf gpurun_out/r04/trace_staged
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_default_s20.json 2> gpurun_out/r04/bench_default_s20.err; cut -c1-400 gpurun_out/r04/bench_default_s20.json
