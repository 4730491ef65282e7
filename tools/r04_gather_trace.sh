cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
python tools/host_issue_probe.py
python tools/host_issue_probe.py --gather
python tools/host_issue_probe.py --gather --timing
rm -rf gpurun_out/r04/trace_gather
timeout 300 rocprofv3 --kernel-trace -f csv -d gpurun_out/r04/trace_gather -o t -- python bench.py --force-gather-path --no-cpu-baseline --no-legs --steps 40 --warmup 5 > gpurun_out/r04/gather_trace.log 2>&1
F=$(find gpurun_out/r04/trace_gather -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $F 1300 40
rm -rf gpurun_out/r04/trace_gather
