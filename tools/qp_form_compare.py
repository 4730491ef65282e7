"""Development aid: the benchmark batch through the whole cycle, outputs saved to an .npz - run once with `path_qp_form=1` (any
`name=value` of emplanner_carla_amd._lib.OPTIONS after the output file) and once without, then
`python tools/qp_form_compare.py diff a.npz b.npz`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if sys.argv[1] == "diff":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    print("status equal:", np.array_equal(a["status"], b["status"]), " differing scenes:", np.nonzero(a["status"] != b["status"])[0][:20])
    ok = ((a["status"] & ~1) == 0) & ((b["status"] & ~1) == 0)
    d = np.abs(a["path_l"] - b["path_l"]).max(axis=1)
    print("scenes planned by both:", int(ok.sum()), " worst |path_l| difference:", d[ok].max(), " scenes beyond 1e-7:", np.nonzero(ok & (d > 1e-7))[0][:40])
    dt = np.abs(a["traj"] - b["traj"]).reshape(len(d), -1).max(axis=1)
    print("worst trajectory difference:", dt[ok].max())
    sys.exit(0)
import torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
cfg, B = S.CFG2, int(os.environ.get("SCENES", "4096"))
batch = S.make_batch(range(B), cfg); P = batch.ref.shape[1]
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
pl = Planner(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    pl.set_option(k, int(v))
r = pl.plan_cycle(p, q, sp, max_pts=max_path_points(p), ref_line=batch.ref, n_ref=np.full(B, P, np.int32), origin_xy=batch.origin_xy,
                  start_xy=batch.start_xy, start_v=batch.start_v, start_a=batch.start_a, obs_xy=batch.obs_xy, n_obs=batch.n_obs)
np.savez(sys.argv[1], status=r.status, path_l=r.path_l, path_s=r.path_s, traj=r.traj, traj_len=r.traj_len)
print("saved", sys.argv[1], "planned", int(((r.status & ~1) == 0).sum()))
