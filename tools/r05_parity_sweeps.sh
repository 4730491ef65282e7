cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05s
python tools/parity_sweep.py 65536 4096 16 0 0 > gpurun_out/r05s/bench_batch.log 2>&1; cp gpurun_out/parity_sweep.json gpurun_out/r05s/bench_batch.json; tail -3 gpurun_out/r05s/bench_batch.log
SWEEP_GEOMETRY=survey python tools/parity_sweep.py 0 4096 16 0 0 > gpurun_out/r05s/tight.log 2>&1; cp gpurun_out/parity_sweep.json gpurun_out/r05s/tight.json; tail -2 gpurun_out/r05s/tight.log
SWEEP_DP_CFG=cfg5 python tools/parity_sweep.py 2048 0 16 0 0 > gpurun_out/r05s/cfg5.log 2>&1; cp gpurun_out/parity_sweep.json gpurun_out/r05s/cfg5.json; tail -2 gpurun_out/r05s/cfg5.log
