#!/bin/bash
# Files what tools/r06_evidence.sh wrote under gpurun_out/ into profiles/ (tracked) and refreshes profiles/counters.json.
cd "$(dirname "$0")/.."
for f in gpurun_out/r06e/bench_*.json; do cp $f profiles/r06_$(basename $f); done
cp gpurun_out/r06e/host_rate_probe.txt profiles/r06_host_rate_probe.txt
cp gpurun_out/r06e/call_cost_probe.txt profiles/r06_call_cost_probe.txt
python tools/summarize_profile.py r06b_32768 32768 30 5 | tail -1
python tools/summarize_profile.py r06c_cfg5 4096 10 3 cfg5_120x21_16obs | tail -1
python tools/summarize_profile.py r06a 4096 20 5 | tail -3          # last: the default workload's entry wins a key collision
python tools/summarize_sq.py gpurun_out/sq_r06c_cfg5 r06c_cfg5 cfg5_120x21_16obs 4096 corridor > /dev/null
python tools/summarize_sq.py gpurun_out/sq_r06a r06a cfg2_40x9_8obs 4096 corridor > /dev/null
git status --short profiles | head -40
