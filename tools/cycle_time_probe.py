"""Per-kernel times of the planning cycle on the bench scenes with a development build of the library
(EMP_DBG_LIB=...); development aid, not the judged bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emplanner_carla_amd import _lib
if os.environ.get("EMP_DBG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["EMP_DBG_LIB"])
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params, max_path_points
cfg = S.CFG2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = S.make_batch(range(B), cfg); P = b.ref.shape[1]
dev = torch.device("cuda:0"); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ins = dict(ref_line=t(b.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(b.origin_xy), start_xy=t(b.start_xy), start_v=t(b.start_v),
           start_a=t(b.start_a), obs_xy=t(b.obs_xy), n_obs=t(b.n_obs))
pl = Planner(0); p = dp_params_from_cfg(cfg); q = qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width); sp = smooth_params()
M = max_path_points(p)
for _ in range(5): r = pl.plan_cycle(p, q, sp, max_pts=M, **ins)
pl.synchronize(); t0 = time.perf_counter()
for _ in range(30): r = pl.plan_cycle(p, q, sp, max_pts=M, **ins)
pl.synchronize(); dt = (time.perf_counter() - t0) / 30
pl.set_timing(True)
for _ in range(5): r = pl.plan_cycle(p, q, sp, max_pts=M, **ins)
pl.synchronize()
st = r.status.cpu().numpy()
print(os.environ.get("EMP_DBG_LIB", "default"), "ms/step %.4f" % (dt * 1e3), "path_qp %.4f" % pl.kernel_ms("path_qp"),
      "to_cartesian %.4f" % pl.kernel_ms("to_cartesian"), "ok %.4f" % ((st & ~1) == 0).mean(), "qp-failed", int(((st & 8) != 0).sum()))
