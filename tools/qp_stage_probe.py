"""Development probe: time of the cycle QP kernel when cut short after successive stages."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params, max_path_points
cfg = S.CFG2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = S.make_batch(range(B), cfg)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
P = batch.ref.shape[1]
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy), start_xy=t(batch.start_xy),
              start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
pl = Planner(0)
p, sp = dp_params_from_cfg(cfg), smooth_params()
for stage, label in ((1, "decimate + lmin/lmax"), (2, "+ QP setup (lane 0)"), (3, "+ initial solve"), (10, "+ IPM init (0 iterations)"),
                     (11, "+ 1 iteration"), (12, "+ 2 iterations"), (14, "+ 4 iterations"), (18, "+ 8 iterations"), (26, "+ 16 iterations"), (0, "full")):
    q = qp_params(reserved=stage)
    for _ in range(3):
        pl.plan_cycle(p, q, sp, max_pts=max_path_points(p), **inputs)
    pl.set_timing(True)
    for _ in range(10):
        pl.plan_cycle(p, q, sp, max_pts=max_path_points(p), **inputs)
    pl.synchronize()
    print(f"stage {stage:2d} {label:28s}: path_qp {pl.kernel_ms('path_qp')*1e3:8.1f} us   to_cartesian {pl.kernel_ms('to_cartesian')*1e3:8.1f} us")
    pl.set_timing(False)
