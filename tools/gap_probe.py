"""Where the GPU idles between the kernels of consecutive staged steps: 60 steps without any timing event, a marker launch
(dp_plan on 8 scenes), 60 steps with the sweep bracketed by HIP events (what bench.py's timed region does).  Run under
`rocprofv3 --kernel-trace` and read the gaps with tools/timeline.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import _lib as L
L.configure_hw_queues(8)
import numpy as np, torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
dev = torch.device("cuda", 0); cfg, B = S.CFG2, 4096
batch = S.make_batch(range(B), cfg); P = batch.ref.shape[1]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy), start_xy=t(batch.start_xy),
              start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p); pl = Planner(0)
pipe = 1                                     # pipeline=N: N lanes instead of the staged form
for kv in sys.argv[1:]:                      # name=value options (emp_set_option) before the pipeline is set up
    k, v = kv.split("=")
    if k == "pipeline": pipe = int(v)
    else: pl.set_option(k, int(v))
pl.set_pipeline(pipe); ts = pl.torch_stream()
def step():
    with torch.cuda.stream(ts):
        return pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
def fence(): pl.synchronize(); torch.cuda.synchronize()
for timing in (False, True):
    for _ in range(100): step()
    fence()
    if timing: pl.set_timing(True, only="dp_sweep")
    t0 = time.perf_counter()
    for _ in range(60): step()
    fence()
    print("sweep events", timing, "%.4f ms per step" % ((time.perf_counter() - t0) / 60 * 1e3), flush=True)
pl.close()
