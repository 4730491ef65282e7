"""Why is a 20-step timed region slower per step than a 100-step one?  Per-round: ms per step, host issue time, first call -
with and without the sweep's timing events.  (Answer: not the host and not event creation - the first rounds after a short
warm-up run at lower clocks; bench.py therefore settles for ~50 ms before it times.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emplanner_carla_amd import _lib as L, scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
dev = torch.device("cuda", 0); cfg, B = S.CFG2, 4096
batch = S.make_batch(range(B), cfg); P = batch.ref.shape[1]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy), start_xy=t(batch.start_xy),
              start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p); pl = Planner(0); pl.set_pipeline(True); ts = pl.torch_stream()
def step():
    with torch.cuda.stream(ts):
        return pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
def fence(): pl.synchronize(); torch.cuda.synchronize()
for mode in ("off", "on-fresh", "on-precreated", "off"):
    for _ in range(5): step()
    fence()
    if mode == "on-fresh": pl.set_timing(True, only="dp_sweep")
    if mode == "on-precreated":
        pl.set_timing(True, only="dp_sweep")   # pairs exist from the previous round
    if mode == "off": pl.set_timing(False)
    t0 = time.perf_counter(); th = []
    for _ in range(20): step(); th.append(time.perf_counter() - t0)
    host = time.perf_counter() - t0
    fence(); el = time.perf_counter() - t0
    print(mode, "20 steps %.4f ms/step, host issue %.3f ms total, first call %.3f ms" % (el / 20 * 1e3, host * 1e3, th[0] * 1e3))
