#!/bin/bash
# Files what tools/r03_evidence.sh wrote under gpurun_out/ into profiles/ (tracked) and refreshes profiles/counters.json.
cd "$(dirname "$0")/.."
for f in gpurun_out/r03/bench_*.json; do cp $f profiles/r03_$(basename $f); done
python tools/summarize_profile.py r03b_32768 32768 30 5 | tail -1
python tools/summarize_profile.py r03c_cfg5 4096 10 3 cfg5_120x21_16obs | tail -1
python tools/summarize_profile.py r03a 4096 100 10 | tail -1          # last: the default workload's entry wins a key collision
python tools/summarize_sq.py gpurun_out/sq_r03c_cfg5 r03c_cfg5 cfg5_120x21_16obs 4096 corridor > /dev/null
python tools/summarize_sq.py gpurun_out/sq_r03a r03a cfg2_40x9_8obs 4096 corridor > /dev/null
git status --short profiles | head -40
