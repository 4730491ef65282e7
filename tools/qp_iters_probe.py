"""Development probe: IPM iteration histogram of the path QP on bench scenes (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emplanner_carla_amd import _lib
if os.environ.get("EMP_DBG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["EMP_DBG_LIB"])
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params, max_path_points
cfg = S.CFG2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = S.make_batch(range(B), cfg)
P = b.ref.shape[1]
pl = Planner(0)
p = dp_params_from_cfg(cfg)
sm, os_, ol_, bsl, start = pl.frenet_project(b.ref, np.full(B, P, np.int32), b.origin_xy, b.start_xy, b.start_v, b.start_a, b.obs_xy, b.n_obs)
rows, mc, st = pl.dp_plan(p, os_, ol_, b.n_obs, start)
M = max_path_points(p)
ps, pll, ln, st2 = pl.dp_enrich(p, rows, start, M)
n = (ln + 1) // 2
dps = np.ascontiguousarray(ps[:, ::2]); dpl = np.ascontiguousarray(pll[:, ::2])
lo, hi, st3 = pl.lmin_lmax(dps, dpl, n.astype(np.int32), os_, ol_, b.n_obs, 5, 5)
l, dl, ddl, iters, st4 = pl.path_qp(qp_params(), lo, hi, n.astype(np.int32), np.ascontiguousarray(start[:, 1:]))
pl.set_timing(True)
for _ in range(5):
    pl.path_qp(qp_params(), lo, hi, n.astype(np.int32), np.ascontiguousarray(start[:, 1:]))
pl.synchronize()
print("path_qp kernel (one scene per wavefront form) %.4f ms" % pl.kernel_ms("path_qp"))
print("status counts", {int(k): int((st4 == k).sum()) for k in np.unique(st4)})
for k in np.unique(st4):
    print("status", k, "iters mean %.1f max %d" % (iters[st4 == k].mean(), iters[st4 == k].max()), np.bincount(iters[st4 == k]))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/qp_cases.npz", lo=lo, hi=hi, n=n.astype(np.int32), start=np.ascontiguousarray(start[:, 1:]), iters=iters, status=st4)
