"""Development probe: iteration counts of the smoothing QP on the trajectories of the bench scenes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params, max_path_points
cfg = S.CFG2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
b = S.make_batch(range(B), cfg)
P = b.ref.shape[1]
pl = Planner(0)
p = dp_params_from_cfg(cfg)
res = pl.plan_cycle(p, qp_params(), smooth_params(), max_pts=max_path_points(p), ref_line=b.ref, n_ref=np.full(B, P, np.int32),
                    origin_xy=b.origin_xy, start_xy=b.start_xy, start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
ok = res.status == 0
print("ok scenes", int(ok.sum()), "traj_len", np.bincount(res.traj_len[ok]))
# re-smooth the raw Cartesian points through the stand-alone entry point to read the iteration counts
sm, os_, ol_, bsl, start = pl.frenet_project(b.ref, np.full(B, P, np.int32), b.origin_xy, b.start_xy, b.start_v, b.start_a, b.obs_xy, b.n_obs)
txy, n_out, st = pl.frenet_path_to_xy(b.ref, sm, np.full(B, P, np.int32), bsl, res.path_s, res.path_l, res.path_len)
out, it, st2 = pl.smooth_line(smooth_params(), txy, n_out)
sel = ok & (st2 == 0)
print("smoothing iterations: mean %.1f max %d" % (it[sel].mean(), it[sel].max()), np.bincount(it[sel]))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/smooth_cases.npz", txy=txy[sel], n=n_out[sel], iters=it[sel])
