"""Which scene is 'batch 1, planned-scene index 436' of tests/test_gpu_fullsize.py::test_pipelined_cycles_equal_plain_cycles
and how many path-QP iterations does it take compared with the rest of its batch?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params, max_path_points
cfg = S.CFG2
k = 1
n_sc = 1536 + 64 * k
b = S.make_batch(range(1000 * k, 1000 * k + n_sc), cfg)
B, P = b.ref.shape[:2]
pl = Planner(0)
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
r = pl.plan_cycle(p, q, sp, ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy,
                  start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
ok = np.nonzero((r.status & ~1) == 0)[0]
print("planned scenes", ok.size, "-> index 436 is scene", ok[436], "of the batch (seed", 1000 * k + ok[436], ")")
i = int(ok[436])
sm, os_, ol_, bsl, start = pl.frenet_project(b.ref, np.full(B, P, np.int32), b.origin_xy, b.start_xy, b.start_v, b.start_a, b.obs_xy, b.n_obs)
rows, mc, st = pl.dp_plan(p, os_, ol_, b.n_obs, start)
M = max_path_points(p)
ps, pll, ln, st2 = pl.dp_enrich(p, rows, start, M)
n = (ln + 1) // 2
dps = np.ascontiguousarray(ps[:, ::2]); dpl = np.ascontiguousarray(pll[:, ::2])
lo, hi, st3 = pl.lmin_lmax(dps, dpl, n.astype(np.int32), os_, ol_, b.n_obs, cfg.obs_length, cfg.obs_width)
l, dl, ddl, iters, st4 = pl.path_qp(q, lo, hi, n.astype(np.int32), np.ascontiguousarray(start[:, 1:]))
print("scene", i, "status", r.status[i], "n", n[i], "n_obs", b.n_obs[i], "stand-alone QP status", st4[i], "iterations", iters[i])
good = st4 == 0
print("iterations over the batch's solved scenes: mean %.1f max %d, histogram" % (iters[good].mean(), iters[good].max()), np.bincount(iters[good]))
print("scenes with the most iterations:", np.argsort(-np.where(good, iters, 0))[:8], np.sort(np.where(good, iters, 0))[::-1][:8])
print("neighbour in the wavefront: scene", i ^ 1, "status", r.status[i ^ 1], "iters", iters[i ^ 1], "QP status", st4[i ^ 1])
print("l_min / l_max of the scene:", lo[i, :n[i]], hi[i, :n[i]])
