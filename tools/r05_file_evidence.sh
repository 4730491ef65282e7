#!/bin/bash
# Files what tools/r05_evidence.sh wrote under gpurun_out/ into profiles/ (tracked) and refreshes profiles/counters.json.
cd "$(dirname "$0")/.."
for f in gpurun_out/r05e/bench_*.json; do cp $f profiles/r05_$(basename $f); done
cp gpurun_out/r05e/host_rate_probe.txt profiles/r05_host_rate_probe.txt
cp gpurun_out/r05e/pcie_probe.txt profiles/r05_pcie_probe.txt
python tools/summarize_profile.py r05b_32768 32768 30 5 | tail -1
python tools/summarize_profile.py r05c_cfg5 4096 10 3 cfg5_120x21_16obs | tail -1
python tools/summarize_profile.py r05a 4096 20 5 | tail -3          # last: the default workload's entry wins a key collision
python tools/summarize_sq.py gpurun_out/sq_r05c_cfg5 r05c_cfg5 cfg5_120x21_16obs 4096 corridor > /dev/null
python tools/summarize_sq.py gpurun_out/sq_r05a r05a cfg2_40x9_8obs 4096 corridor > /dev/null
git status --short profiles | head -40
