import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params, max_path_points
cfg = S.CFG2; B = 4096
b = S.make_batch(range(B), cfg); P = b.ref.shape[1]
dev = torch.device("cuda:0"); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ins = dict(ref_line=t(b.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(b.origin_xy), start_xy=t(b.start_xy), start_v=t(b.start_v), start_a=t(b.start_a), obs_xy=t(b.obs_xy), n_obs=t(b.n_obs))
pl = Planner(0); p = dp_params_from_cfg(cfg); q = qp_params(); sp = smooth_params(); M = max_path_points(p)
for timing in (False, True, False, True):
    pl.set_timing(timing)
    for _ in range(5): pl.plan_cycle(p, q, sp, max_pts=M, **ins)
    pl.synchronize(); t0 = time.perf_counter()
    for _ in range(50): pl.plan_cycle(p, q, sp, max_pts=M, **ins)
    pl.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("timing", timing, "ms/step %.4f" % (dt * 1e3))
