import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params, max_path_points
from oracle import ref_port as op
cfg = S.LatticeConfig("long_100x13", row=13, col=100, sample_s=1.1, sample_l=0.9, sampling_res=1, n_obs=10, n_ref=75)
seeds = [5, 6, 7, 8]
b = S.make_batch(seeds, cfg); P = b.ref.shape[1]
pl = Planner(0); p = dp_params_from_cfg(cfg)
r = pl.plan_cycle(p, qp_params(), smooth_params(), max_pts=max_path_points(p), ref_line=b.ref, n_ref=np.full(len(seeds), P, np.int32),
                  origin_xy=b.origin_xy, start_xy=b.start_xy, start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
print("device status", r.status, "traj_len", r.traj_len, "path_len", r.path_len)
for k in range(len(seeds)):
    t0 = time.time()
    try:
        out = op.plan_cycle(b.ref[k], tuple(b.origin_xy[k]), tuple(b.start_xy[k]), tuple(b.start_v[k]), tuple(b.start_a[k]),
                            [tuple(o) for o in b.obs_xy[k, :int(b.n_obs[k])]],
                            dp_kwargs=dict(row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l, sampling_res=cfg.sampling_res), verbose=False)
        ok = out.get("qp_status") == "optimal" and out["smooth_status"] == "optimal"
        want = np.asarray(out["trajectory"]); m = int(r.traj_len[k])
        err = np.abs(r.traj[k, :m, :2] - want[:m, :2]).max() if ok and m == len(want) else None
        print(k, "port ok", ok, "dp feasible", out["dp_feasible"], "len", len(want), "max |dxy|", err, "%.0f s" % (time.time() - t0))
    except IndexError as e:
        print(k, "port IndexError", e)
