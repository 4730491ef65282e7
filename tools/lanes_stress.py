"""bench.py's timed loop, checked: steps issued back to back on the planner's stream (no host wait in between), rotating through
four resident input batches of 4096 scenes, EVERY step's outputs compared with the plain (one batch at a time) result of its
batch - for three lanes (with and without EMP_OPT_LANE_EDGE_ORDER), six lanes and the staged form.
Usage: python tools/lanes_stress.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import _lib as L
L.configure_hw_queues(12)
import numpy as np
import torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg, B = S.CFG2, 4096
dev = torch.device("cuda:0")
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ring = []
for k in range(4):
    b = S.make_batch(range(k * B, (k + 1) * B), cfg, start_ahead=S.BENCH_START_AHEAD)
    ring.append(dict(ref_line=t(b.ref), n_ref=t(np.full(B, b.ref.shape[1], np.int32)), origin_xy=t(b.origin_xy), start_xy=t(b.start_xy),
                     start_v=t(b.start_v), start_a=t(b.start_a), obs_xy=t(b.obs_xy), n_obs=t(b.n_obs)))
FIELDS = ("dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len", "traj", "traj_len", "status")
pl = Planner(0)
plain = []
for ins in ring:
    r = pl.plan_cycle(p, q, sp, max_pts=M, **ins)
    pl.synchronize()
    plain.append({k: getattr(r, k).clone() for k in FIELDS})
bad = 0
for label, pipe, opts in (("3 lanes", 3, {}), ("3 lanes, lane_edge_order=1", 3, {"lane_edge_order": 1}), ("6 lanes", 6, {}), ("staged", "staged", {})):
    for k, v in opts.items():
        pl.set_option(k, v)
    pl.set_pipeline(pipe)
    depth = pl.in_flight if pipe != "staged" else 2
    ts = pl.torch_stream()
    steps = 0
    for rnd in range(R):
        window = []
        burst = depth * (1 + rnd % 7)               # up to seven pipeline depths back to back; the last `depth` results are still
        for i in range(burst):                      # referenced by the pipeline (older outputs went back to torch's allocator)
            ib = (rnd * 5 + i) % 4
            with torch.cuda.stream(ts):
                r_i = pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **ring[ib])
            window.append((ib, r_i))
            if len(window) > depth:
                window.pop(0)
        pl.synchronize(); torch.cuda.synchronize()
        for ib, r in window:
            steps += 1
            for k in FIELDS:
                if not torch.equal(getattr(r, k), plain[ib][k]) and not (k in ("dp_s", "dp_l", "path_s", "path_l", "traj", "dp_rows") and
                                                                       torch.equal(torch.nan_to_num(getattr(r, k)), torch.nan_to_num(plain[ib][k]))):
                    bad += 1
                    print(f"{label}: round {rnd} batch {ib} field {k} differs", flush=True)
    pl.set_pipeline(False)
    for k in opts:
        pl.set_option(k, 0)
    print(f"{label}: {steps} steps checked, mismatching fields so far {bad}", flush=True)
pl.close()
print("LANES-STRESS", "OK" if bad == 0 else f"FAILED ({bad})")
