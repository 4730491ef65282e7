# Round-6 parity sweeps on the GPU box (the QP kernels' arithmetic changed: fused multiply-adds inside the interior-point code):
# the benchmark batch, SURVEY 8(d)'s tight arcs, the on-node batch of rounds 1-4 (last-bit-sensitive starts against the
# port), configs[4]'s DP against the exact oracle.  tools/parity_sweep.py writes gpurun_out/parity_sweep.json; copies under gpurun_out/r06s.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06s
python tools/parity_sweep.py 65536 4096 16 0 0 > gpurun_out/r06s/bench_batch.log 2>&1; cp gpurun_out/parity_sweep.json gpurun_out/r06s/bench_batch.json; tail -3 gpurun_out/r06s/bench_batch.log
SWEEP_GEOMETRY=survey python tools/parity_sweep.py 0 4096 16 0 0 > gpurun_out/r06s/tight.log 2>&1; cp gpurun_out/parity_sweep.json gpurun_out/r06s/tight.json; tail -2 gpurun_out/r06s/tight.log
SWEEP_START_AHEAD=2.0 python tools/parity_sweep.py 16384 2048 16 0 0 > gpurun_out/r06s/on_node.log 2>&1; cp gpurun_out/parity_sweep.json gpurun_out/r06s/on_node.json; tail -3 gpurun_out/r06s/on_node.log
SWEEP_DP_CFG=cfg5 python tools/parity_sweep.py 2048 0 16 0 0 > gpurun_out/r06s/cfg5.log 2>&1; cp gpurun_out/parity_sweep.json gpurun_out/r06s/cfg5.json; tail -2 gpurun_out/r06s/cfg5.log
python tools/parity_sweep.py 0 0 16 8192 4096 > gpurun_out/r06s/st_fe.log 2>&1; cp gpurun_out/parity_sweep.json gpurun_out/r06s/st_fe.json; tail -3 gpurun_out/r06s/st_fe.log
